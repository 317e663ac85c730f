"""Measure-only (round-4 review, item 10): would an overlap-save FFT form of the matched filter beat the MFMA
kernel?  In that form every (template, channel, block of 4096 - L + 1 lags) costs ONE 4096-point complex-to-real
inverse transform (the data blocks' forward transforms are shared by all templates); the question the review put:
is hipFFT's batched 4096-point C2R rate at least 3x the lag rate of mf_mfma_wave_kernel?  That kernel computes
T x 60 x 8 639 745 channel-lags in ~0.98 s = 2.65e11 channel-lags/s; a C2R transform delivers 3841 of them
(L = 256), so 3x needs 2.07e8 transforms/s = 6.6 TB/s of HBM traffic at 32 KB per transform."""
import time

import torch

dev = torch.device("cuda")
n, L = 4096, 256
valid = n - L + 1
mfma_rate = 500 * 60 * 8_639_745 / 0.98
for batch in (16_384, 65_536, 262_144):
    spec = torch.randn((batch, n // 2 + 1), dtype=torch.complex64, device=dev)
    tmpl = torch.randn((batch, n // 2 + 1), dtype=torch.complex64, device=dev)
    out = torch.fft.irfft(spec, n=n, dim=-1)
    torch.cuda.synchronize()
    for label, fn in (("C2R alone", lambda: torch.fft.irfft(spec, n=n, dim=-1)),
                      ("multiply by the template spectrum + C2R", lambda: torch.fft.irfft(spec * tmpl, n=n, dim=-1))):
        best = 1e9
        for _ in range(5):
            t0 = time.perf_counter()
            for _ in range(4):
                out = fn()
            torch.cuda.synchronize()
            best = min(best, (time.perf_counter() - t0) / 4)
        rate = batch / best
        print(f"batch {batch:7d}, {label:42s}: {rate / 1e6:8.1f} M transforms/s = {rate * valid / 1e11:6.2f}e11 channel-lags/s "
              f"= {rate * valid / mfma_rate:5.2f} x the MFMA kernel ({rate * 32768 / 1e12:.2f} TB/s of transform traffic)")
    del spec, tmpl, out
