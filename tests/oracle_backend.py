"""Scenario back-end over the CPU oracle."""
from oracle import cpu


def seg_from_dict(d):
  return cpu.seg(d["dim"], d.get("opt", "sgd"), d.get("init", "zeros"),
                 init_acc=d.get("init_acc", 0.1), weight_decay=d.get("weight_decay", 0.0),
                 l1=d.get("l1", 0.0), l2=d.get("l2", 0.0), beta=d.get("beta", 0.0),
                 init_a=d.get("init_a", 0.0), init_b=d.get("init_b", 0.0),
                 momentum=d.get("momentum", 0.9), epsilon=d.get("epsilon", 0.01),
                 averaging_ratio=d.get("averaging_ratio", 0.9),
                 learning_rate=d.get("learning_rate", 0.01),
                 use_nesterov=d.get("use_nesterov", False), beta1=d.get("beta1", 0.9),
                 beta2=d.get("beta2", 0.99), comp=d.get("comp", "fp32"),
                 comp_a=d.get("comp_a", 0.0), comp_b=d.get("comp_b", 1.0),
                 stochastic_rounding_float16=bool(d.get("sr16")))


def make_table(cfg):
  return cpu.OracleTable([seg_from_dict(s) for s in cfg["segments"]],
                         initial_capacity=cfg.get("initial_capacity", 1),
                         default_expire_days=cfg.get("default_expire_days", 36500),
                         slot_expire={int(k): v for k, v in cfg.get("slot_expire", {}).items()},
                         seed=cfg.get("seed", 0), serving=bool(cfg.get("serving")))


def make_filter(cfg):
  return cpu.OracleFilter(cfg.get("kind", "sliding"), cfg.get("capacity", 300000000),
                          cfg.get("split_num", 7), cfg.get("default_threshold", 0),
                          {int(k): v for k, v in cfg.get("slot_thresholds", {}).items()})
