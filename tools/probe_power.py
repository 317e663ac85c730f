"""Socket power and shader clock (rocm-smi, sampled twice a second) while one of the two hot kernels runs back to back:
is the clock the BP kernel sustains (1.9 - 2.1 GHz of 2.4) a power cap?  usage: python tools/probe_power.py mf|mf_split16|bp [seconds]"""
import os, subprocess, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
import seismic_bpmf_amd as sb
from seismic_bpmf_amd import synthetic as syn

which = sys.argv[1] if len(sys.argv) > 1 else "bp"
seconds = float(sys.argv[2]) if len(sys.argv) > 2 else 12.0
device = torch.device("cuda", 0)
if which in ("mf", "mf_split16"):
    if which == "mf_split16":
        sb.set_option("mf.split16", 1)
    cfg = dict(syn.MF_CONFIGS["cfg2"]); cfg["T"] = 100
    tmpl, mv, w, data, _ = bench.mf_inputs_device(cfg, device, 20260928, 0)
    mf = sb.MatchedFilterGPU(device=0); mf.set_data(data)
    cc = torch.empty((cfg["T"], cfg["N"] - cfg["L"] + 1), dtype=torch.float32, device=device)
    step = lambda: mf.run(tmpl, mv, w, 1, out=cc)
else:
    bcfg = dict(syn.BP_CONFIGS["cfg3"])
    geo, feat, wp = bench.bp_inputs(bcfg, device, 20260928, 0, 1)
    bf = sb.BeamformerGPU(geo["moveouts"], geo["weights_sources"], device=0)
    beam = torch.empty(bcfg["N"], dtype=torch.float32, device=device)
    arg = torch.empty(bcfg["N"], dtype=torch.int32, device=device)
    step = lambda: bf.run(feat, wp, out=(beam, arg))
step(); torch.cuda.synchronize()
samples, stop = [], False
def sampler():
    while not stop:
        out = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--showtemp"], capture_output=True, text=True).stdout
        samples.append((time.perf_counter(), out))
        time.sleep(0.4)
th = threading.Thread(target=sampler); th.start()
t0 = time.perf_counter(); n = 0
while time.perf_counter() - t0 < seconds:
    step(); n += 1
    if n % 4 == 0:
        torch.cuda.synchronize()
torch.cuda.synchronize()
dt = time.perf_counter() - t0
stop = True; th.join()
print(f"{which}: {n} steps in {dt:.2f} s = {dt / n * 1e3:.1f} ms per step")
import re
for t, out in samples[2:-1:3]:
    pw = re.findall(r"(?:Power|power)[^\n]*?:\s*([0-9.]+)", out)
    sclk = re.findall(r"sclk clock level[^\n]*\((\d+)Mhz\)", out)
    temp = re.findall(r"Temperature \(Sensor (?:junction|hotspot)[^\n]*?:\s*([0-9.]+)", out)
    print(f"  t={t - t0:5.1f}s power {pw[:2]} W  sclk {sclk[:1]} MHz  temp {temp[:1]}")
print(samples[len(samples) // 2][1][:1500])
