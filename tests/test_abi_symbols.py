"""CPU-only: the C-ABI library loads and exports every symbol include/monolith_hip.h declares."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
  text = open(os.path.join(ROOT, "include", "monolith_hip.h")).read()
  text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
  return sorted(set(re.findall(r"\b(mono_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
  names = _declared()
  assert len(names) > 30
  lib = ctypes.CDLL(os.path.join(ROOT, "monolith_amd", "libmonolith_hip.so"))
  missing = [n for n in names if not hasattr(lib, n)]
  assert not missing, "declared in include/monolith_hip.h but not exported: %s" % missing
  assert lib.mono_abi_version() >= 1


def test_product_does_not_touch_the_oracle():
  """The product path must not import / link anything under oracle/."""
  for dirpath, _, files in os.walk(os.path.join(ROOT, "monolith_amd")):
    for f in files:
      if f.endswith((".py", ".hip", ".h", ".cpp")):
        src = open(os.path.join(dirpath, f)).read()
        assert "liboracle" not in src and "from oracle" not in src and "import oracle" not in src, f
