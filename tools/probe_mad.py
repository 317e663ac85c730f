"""MAD detection threshold of a whole cfg2-sized CC matrix (500 rows x 8.64 M samples, 30-min windows,
overlap 0.25 as in the workflow) on the device: time of bpmf_tdt_mad_dev; and the row kurtosis."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from seismic_bpmf_amd.threshold import ThresholdGPU
from seismic_bpmf_amd import workflow
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 500
n = 8_640_000 - 255
g = torch.Generator(device="cuda"); g.manual_seed(1)
cc = torch.randn((rows, n), device="cuda", generator=g) * 0.05
cc[:, :1500] = 0.0
cc[:, -1500:] = 0.0
th = ThresholdGPU()
wn = np.random.default_rng(0).standard_normal(10_000).astype(np.float32)
for ov in (0.25, 0.66):
    W = 180_000
    thr, _ = th.time_dependent_threshold_mad(cc, W, 8.0, overlap=ov, white_noise=wn, expand=False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    thr, _ = th.time_dependent_threshold_mad(cc, W, 8.0, overlap=ov, white_noise=wn, expand=False)
    torch.cuda.synchronize()
    print(f"MAD threshold, {rows} rows x {n}, window {W}, overlap {ov}: {(time.perf_counter() - t0) * 1e3:.1f} ms ({thr.shape[1]} windows/row)")
del th
torch.cuda.empty_cache()
k = workflow.row_excess_kurtosis(cc)
torch.cuda.synchronize()
t0 = time.perf_counter()
k = workflow.row_excess_kurtosis(cc)
print(f"row kurtosis of {rows} rows: {(time.perf_counter() - t0) * 1e3:.1f} ms; mean {np.nanmean(k):.3f}")
