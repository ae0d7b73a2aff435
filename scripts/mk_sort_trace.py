"""Instrumented copy of sort.hip for scripts/exp_sort_trace.py: stamps wall_clock64 after the load
and after every __syncthreads of k_sort_fused into a device array. Builds
scripts/bin/libmonolith_hip_trace.so from a scratch tree under /tmp; the product is not touched."""
import os, re, shutil, subprocess, sys
KERNEL = sys.argv[1] if len(sys.argv) > 1 else "k_sort_fused"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
dst = "/tmp/sort_trace"
shutil.rmtree(dst, ignore_errors=True)
os.makedirs(dst + "/monolith_amd/csrc"); os.makedirs(dst + "/include")
for f in os.listdir(root + "/monolith_amd/csrc"):
  if f.endswith((".hip", ".h")) or f == "Makefile":
    shutil.copy(root + "/monolith_amd/csrc/" + f, dst + "/monolith_amd/csrc/")
for f in os.listdir(root + "/include"):
  shutil.copy(root + "/include/" + f, dst + "/include/")
p = dst + "/monolith_amd/csrc/sort.hip"
s = open(p).read()
s = s.replace("namespace mono {\n", "namespace mono {\n__device__ unsigned long long g_trace[512 * 16];\n", 1)
a = s.index("__global__ void __launch_bounds__(kSortThreads) %s(" % KERNEL)
b = s.index("\n}\n", a) + 3
body = s[a:b]
ld = "    val[r] = valid ? (vin ? vin[i] : (uint32_t)i) : 0u;\n  }\n"
body = body.replace(ld, ld + "  if (threadIdx.x == 0 && (key[15] ^ val[15]) != 0xdeadbeefu) g_trace[blockIdx.x * 16 + 1] = wall_clock64();\n", 1)
k = [0]
def stamp(m):
  k[0] += 1
  idx = {1: 0}.get(k[0], k[0])
  return m.group(0) + "\n  if (threadIdx.x == 0) g_trace[blockIdx.x * 16 + %d] = wall_clock64();" % idx
body = re.sub(r"__syncthreads\(\);", stamp, body)
e = body.rindex("}\n")
body = body[:e] + "  if (threadIdx.x == 0) { g_trace[blockIdx.x * 16 + %d] = wall_clock64(); g_trace[blockIdx.x * 16 + 15] = s_tile; }\n" % (k[0] + 1) + body[e:]
s = s[:a] + body + s[b:]
s += '''
extern "C" int mono_debug_sort_trace(unsigned long long* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(mono::g_trace), sizeof(unsigned long long) * 512 * 16);
}
'''
open(p, "w").write(s)
print("stamps:", k[0] + 2)
mk = dst + "/monolith_amd/csrc/Makefile"
t = open(mk).read().replace("../libmonolith_hip.so", "libmonolith_hip_trace.so")
open(mk, "w").write(t)
subprocess.check_call(["make", "-C", dst + "/monolith_amd/csrc", "-j8"], stdout=subprocess.DEVNULL)
os.makedirs(root + "/scripts/bin", exist_ok=True)
shutil.copy(dst + "/monolith_amd/csrc/libmonolith_hip_trace.so", root + "/scripts/bin/")
