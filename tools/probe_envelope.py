"""Probe: cost of the envelope stage (BPMF/template_search.py:1525-1617) at configs[2] size, 60 channels x
4 320 000 samples, and of the alternatives: the float64 C2C pair of round 3, a float64 R2C + C2R pair
(envelope = sqrt(x^2 + H[x]^2), H[x] = irfft(-i X)), the same in float32, different channel batches."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
from seismic_bpmf_amd import features  # noqa: E402

S, C, N = 20, 3, int(sys.argv[1]) if len(sys.argv) > 1 else 4_320_000
g = torch.Generator(device="cuda")
g.manual_seed(3)
x = torch.randn((S * C, N), device="cuda", generator=g)


def timed(fn, reps=3):
    fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter()
        out = fn()
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    return best * 1e3, out


def c2c(batch):
    return lambda: features.envelope_c2c(x, channels_per_batch=batch)


def r2c(dtype, batch):
    def run():
        out = torch.empty_like(x)
        for i in range(0, x.shape[0], batch):
            xb = x[i:i + batch].to(dtype)
            X = torch.fft.rfft(xb, dim=-1)
            X[:, 0] = 0
            if N % 2 == 0:
                X[:, -1] = 0
            Y = torch.complex(X.imag, -X.real)            # -i X
            h = torch.fft.irfft(Y, n=N, dim=-1)
            out[i:i + batch] = torch.sqrt(xb * xb + h * h).to(torch.float32)
        return out
    return run


ref_ms, ref = timed(c2c(16))
print(f"C2C float64, 16 channels per batch (round 3): {ref_ms:8.1f} ms")
scale = ref.abs().amax(dim=1, keepdim=True)
for name, fn in (("C2C float64, 4 per batch", c2c(4)), ("R2C/C2R float64, 16 per batch", r2c(torch.float64, 16)),
                 ("R2C/C2R float64, 60 per batch", r2c(torch.float64, 60)), ("R2C/C2R float32, 60 per batch", r2c(torch.float32, 60))):
    ms, out = timed(fn)
    err = ((out - ref).abs() / (scale * 2.0 ** -23)).max().item()
    print(f"{name:34s}: {ms:8.1f} ms   max |diff| to the C2C float64 envelope = {err:.2f} ulp of the channel maximum, "
          f"{(out != ref).float().mean().item() * 100:.4f} % of the samples differ")
ms, (feat, avail) = timed(lambda: features.saturated_envelopes(x.reshape(S, C, N)))
print(f"saturated_envelopes (envelope + median / MAD + standardise + clip): {ms:8.1f} ms")
