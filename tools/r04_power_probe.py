"""What can a non-root process read about power / clocks on the GPU box, and how fast does it update?  Runs ~3 s of bench steps
while a side thread samples hwmon + pp_dpm_sclk + gpu_metrics every 2 ms; compares with rocm_smi_lib's energy counter."""
import ctypes, glob, os, struct, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import msnets_amd
from msnets_amd import cbmv_generator, synthetic
from msnets_amd.gcnet_3dcnn import GCNet_CostVolumeAggre

card = [c for c in sorted(glob.glob("/sys/class/drm/card[0-9]*/device")) if os.path.exists(c + "/pp_dpm_sclk")][0]
hw = glob.glob(card + "/hwmon/hwmon*")[0]
def rd(p, mode="r"):
    try:
        with open(p, mode) as f: return f.read()
    except Exception as e: return None
gm = rd(card + "/gpu_metrics", "rb")
print("gpu_metrics bytes", None if gm is None else len(gm), None if gm is None else struct.unpack_from("<HBB", gm, 0))
if gm: print("first 96 bytes:", gm[:96].hex())
rsmi = None
for cand in ("/opt/rocm/lib/librocm_smi64.so", "librocm_smi64.so"):
    try:
        rsmi = ctypes.CDLL(cand); break
    except OSError: pass
def energy():
    if rsmi is None: return None
    e, res, ts = ctypes.c_uint64(), ctypes.c_float(), ctypes.c_uint64()
    rc = rsmi.rsmi_dev_energy_count_get(0, ctypes.byref(e), ctypes.byref(res), ctypes.byref(ts))
    return (rc, e.value, res.value, ts.value)
def rsmi_power():
    p = ctypes.c_uint64(); t = ctypes.c_int()
    rc = rsmi.rsmi_dev_power_get(0, ctypes.byref(p), ctypes.byref(t)) if hasattr(rsmi, "rsmi_dev_power_get") else -1
    return rc, p.value, t.value
if rsmi is not None:
    print("rsmi_init", rsmi.rsmi_init(0), "energy", energy(), "power", rsmi_power())

dev = torch.device("cuda:0")
l, r, _ = synthetic.stereo_pair(272, 480, 96, seed=0)
l, r = torch.from_numpy(l).to(dev), torch.from_numpy(r).to(dev)
torch.manual_seed(0)
model = GCNet_CostVolumeAggre(192).eval().to(dev)
vb = cbmv_generator.VolumeBuilder(292, 500, 96, dev, layout="ndhwc")
vol = torch.empty((1,) + vb.out_shape, device=dev)
def step():
    vb(l, r, out=vol[0]); return model.forward_ndhwc(vol)
for _ in range(5): step()
torch.cuda.synchronize()
samples, stop = [], [False]
def loop():
    while not stop[0]:
        t = time.perf_counter()
        g = rd(card + "/gpu_metrics", "rb")
        samples.append((t, rd(hw + "/power1_input"), rd(hw + "/freq1_input"), [x for x in (rd(card + "/pp_dpm_sclk") or "").splitlines() if x.endswith("*")], g[:64] if g else None, time.perf_counter() - t))
        time.sleep(0.002)
th = threading.Thread(target=loop, daemon=True); th.start()
time.sleep(0.3)
e0 = energy(); t0 = time.perf_counter()
for _ in range(400): step()
torch.cuda.synchronize()
t1 = time.perf_counter(); e1 = energy()
time.sleep(0.3); stop[0] = True; th.join()
print("400 steps in %.3f s = %.3f ms/step" % (t1 - t0, 1e3 * (t1 - t0) / 400))
if e0 and e1 and e0[0] == 0:
    print("rsmi energy: %s -> %s ; delta*res = %.3f J -> %.1f W" % (e0, e1, (e1[1] - e0[1]) * e0[2] * 1e-6, (e1[1] - e0[1]) * e0[2] * 1e-6 / (t1 - t0)))
print("samples", len(samples), "mean read cost %.3f ms" % (1e3 * sum(s[5] for s in samples) / len(samples)))
for s in samples[::max(1, len(samples) // 60)]:
    g = s[4]
    extra = ""
    if g:
        extra = " sockpw(off10)=%d" % struct.unpack_from("<H", g, 10)[0]
    print("%.3f pw %s f %s dpm %s%s" % (s[0] - t0, (s[1] or "").strip(), (s[2] or "").strip(), s[3], extra))
os.system("rocm-smi --showpower --showclocks 2>&1 | grep -E 'Power|sclk' | head -4")
