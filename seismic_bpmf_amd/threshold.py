"""Device-side post-CC step: RMS time-dependent threshold + candidate extraction.

Mirrors ``BPMF.clib.time_dependent_threshold`` (BPMF/clib.py:257-309, RMS variant of
BPMF/libc.c:516-673) for every row of a CC matrix that is already in HBM (the output of
:class:`MatchedFilterGPU`), and the first test of ``MatchedFilter.select_cc_indexes``
(``cc > threshold``, BPMF/similarity_search.py:231-232, capped as in :629).  Only the candidate
records come back to the host, where :func:`seismic_bpmf_amd.postprocess.select_cc_indexes`-style
merging runs on a few thousand entries instead of millions of samples.
"""
import ctypes as C

import numpy as np

from . import _lib

GAUSSIAN_SAMPLE_LEN = 500

candidate_dtype = np.dtype([("row", np.int32), ("index", np.int32), ("cc", np.float32),
                            ("threshold", np.float32)])


def window_params(sliding_window_samp, overlap):
    """(half_window, shift) exactly as BPMF/clib.py:292-293."""
    return int(sliding_window_samp) // 2, int((1.0 - overlap) * int(sliding_window_samp))


class ThresholdGPU:
    def __init__(self, device=None):
        import torch
        self.torch = torch
        if not torch.cuda.is_available():
            raise _lib.BpmfHipError("ThresholdGPU needs a HIP device")
        self.device = torch.device("cuda", torch.cuda.current_device() if device is None else device)
        self.lib = _lib.lib()
        self._ws = None
        self.mad_workspace_limit = 4 << 30      # bytes of workspace per call of the MAD threshold (the rank tables: n / 32 bytes per row)

    def _stream(self):
        return C.c_void_p(self.torch.cuda.current_stream(self.device).cuda_stream)

    def time_dependent_threshold(self, cc, sliding_window_samp, num_dev, overlap=0.66,
                                 white_noise=None, expand=False):
        """cc: (rows, n) float32 device tensor.  Returns (thr_windows (rows, n_win), full or None)."""
        t = self.torch
        cc = cc.contiguous()
        if cc.dim() == 1:
            cc = cc[None]
        rows, n = cc.shape
        half, shift = window_params(sliding_window_samp, overlap)
        n_win = self.lib.bpmf_tdt_num_windows(n, half, shift)
        if n_win == 0:
            raise ValueError("series shorter than the sliding window")
        if white_noise is None:
            white_noise = np.random.normal(size=GAUSSIAN_SAMPLE_LEN).astype("float32")
        g = t.as_tensor(np.ascontiguousarray(white_noise[:GAUSSIAN_SAMPLE_LEN], dtype=np.float32),
                        device=self.device)
        nbytes = self.lib.bpmf_tdt_workspace_bytes(rows, n, half, shift)
        if self._ws is None or self._ws.numel() < nbytes:
            self._ws = t.empty(nbytes, dtype=t.uint8, device=self.device)
        thr_win = t.empty((rows, n_win), dtype=t.float32, device=self.device)
        full = t.empty((rows, n), dtype=t.float32, device=self.device) if expand else None
        with t.cuda.device(self.device):
            rc = self.lib.bpmf_tdt_rms_dev(cc.data_ptr(), g.data_ptr(), float(num_dev), rows, n, half,
                                           shift, self._ws.data_ptr(), self._ws.numel(),
                                           self._stream(), thr_win.data_ptr(),
                                           full.data_ptr() if expand else None)
        _lib.check(rc, "bpmf_tdt_rms_dev")
        self._keep = (cc, g)
        return thr_win, full

    def time_dependent_threshold_mad(self, cc, sliding_window_samp, num_dev, overlap=0.66,
                                     white_noise=None, expand=True):
        """MAD variant of the detection threshold (BPMF/similarity_search.py:1079-1113) for every row
        of a CC matrix that lives on the device, in one call (csrc/stats.hip: radix-select medians,
        every float32 operation in NumPy's order -- equal to postprocess.time_dependent_threshold_mad
        bit for bit).  `white_noise`: standard-normal values; row r fills its zeros with
        white_noise[:n_zeros(r)] (drawn here when None).

        cc (n,) -> the (n,) float32 threshold as a device tensor (the reference's return value);
        cc (rows, n) -> (thr_windows (rows, n_windows), full (rows, n) or None if not `expand`)."""
        t = self.torch
        single = cc.dim() == 1
        x = cc.reshape(1, -1) if single else cc
        x = x.to(device=self.device, dtype=t.float32).contiguous()
        rows, n = x.shape
        W = int(sliding_window_samp)
        shift = int((1.0 - overlap) * W)
        n_win = self.lib.bpmf_tdt_mad_num_windows(n, W, shift)
        if n_win == 0:
            raise ValueError("series shorter than the sliding window")
        if white_noise is None:
            white_noise = np.random.normal(size=n).astype("float32")
        wn = t.as_tensor(np.ascontiguousarray(white_noise, dtype=np.float32), device=self.device)
        # rows go through the library in chunks: at most 65535 per call (its gridDim.y), and few enough that the
        # workspace -- since round 5 only the zero-rank tables (n / 32 bytes per row: the zeros are replaced where
        # the medians read them, there is no filled copy of the matrix any more) and the buffers the row statistics
        # collect the middle of a row in (n / 10 bytes per row) -- stays under
        # `mad_workspace_limit` bytes (4 GiB by default; one row always goes)
        # (per-row bytes from the library itself: short rows take the one-workgroup statistics and need next to nothing,
        # long ones ~n / 6 + 100 KB)
        per_row = max(1, int(self.lib.bpmf_tdt_mad_workspace_bytes(64, n, W, shift)) // 64)
        chunk = int(max(1, min(rows, 65535, self.mad_workspace_limit // per_row)))
        nbytes = self.lib.bpmf_tdt_mad_workspace_bytes(chunk, n, W, shift)
        if self._ws is None or self._ws.numel() < nbytes:
            self._ws = None
            self._ws = t.empty(nbytes, dtype=t.uint8, device=self.device)
        thr_win = t.empty((rows, n_win), dtype=t.float32, device=self.device)
        full = t.empty((rows, n), dtype=t.float32, device=self.device) if (expand or single) else None
        with t.cuda.device(self.device):
            for r0 in range(0, rows, chunk):
                nr = min(chunk, rows - r0)
                rc = self.lib.bpmf_tdt_mad_dev(x[r0:].data_ptr(), wn.data_ptr() if wn.numel() else None, wn.numel(),
                                               float(num_dev), nr, n, W, shift, self._ws.data_ptr(),
                                               self._ws.numel(), self._stream(), thr_win[r0:].data_ptr(),
                                               full[r0:].data_ptr() if full is not None else None, None)
                _lib.check(rc, "bpmf_tdt_mad_dev")
        self._keep = (x, wn)
        return full[0] if single else (thr_win, full)

    def extract_candidates(self, cc, thr_windows, sliding_window_samp, overlap=0.66, row_cap=None,
                           capacity=1 << 20, kind="rms"):
        """Records (row, index, cc, threshold) of every sample above min(threshold, row_cap).
        `thr_windows` are the window values of time_dependent_threshold (kind "rms") or of
        time_dependent_threshold_mad (kind "mad").  The record buffer grows and the extraction is
        repeated when it overflows."""
        t = self.torch
        cc = cc.contiguous()
        if cc.dim() == 1:
            cc = cc[None]
        rows, n = cc.shape
        half, shift = window_params(sliding_window_samp, overlap)
        cap = None
        if row_cap is not None:
            cap = t.as_tensor(np.ascontiguousarray(row_cap, dtype=np.float32), device=self.device)
        thr_windows = thr_windows.contiguous()
        while True:
            count = t.zeros(1, dtype=t.int32, device=self.device)
            rec = t.empty((capacity, 4), dtype=t.int32, device=self.device)
            with t.cuda.device(self.device):
                if kind == "rms":
                    rc = self.lib.bpmf_extract_candidates_dev(
                        cc.data_ptr(), thr_windows.data_ptr(), cap.data_ptr() if cap is not None else None,
                        rows, n, half, shift, capacity, self._stream(), count.data_ptr(), rec.data_ptr())
                elif kind == "mad":
                    rc = self.lib.bpmf_extract_candidates_mad_dev(
                        cc.data_ptr(), thr_windows.data_ptr(), cap.data_ptr() if cap is not None else None,
                        rows, n, int(sliding_window_samp), shift, capacity, self._stream(), count.data_ptr(),
                        rec.data_ptr())
                else:
                    raise ValueError("kind must be 'rms' or 'mad'")
            _lib.check(rc, "bpmf_extract_candidates_dev")
            n_found = int(count.item())
            if n_found <= capacity:
                break
            capacity = n_found
        out = rec[:n_found].cpu().numpy().view(candidate_dtype).reshape(-1)
        return out[np.lexsort((out["index"], out["row"]))]


peak_dtype = np.dtype([("index", np.int32), ("beam", np.float32), ("source", np.int32),
                       ("pad", np.int32)])


class BeamDetectorGPU:
    """Device side of ``Beamformer.find_detections`` (BPMF/template_search.py:574-627): the sliding
    median/MAD statistics of its threshold (:1418-1487) and the extraction of the rising-edge local
    maxima of the max-beam (BPMF/utils.py:2292-2301) above a floor, as compact records.  The (N,)
    max-beam stays in HBM; see csrc/bp_detect.hip and workflow.backprojection_detections."""

    def __init__(self, device=None):
        import torch
        self.torch = torch
        if not torch.cuda.is_available():
            raise _lib.BpmfHipError("BeamDetectorGPU needs a HIP device")
        self.device = torch.device("cuda", torch.cuda.current_device() if device is None else device)
        self.lib = _lib.lib()

    def _stream(self):
        return C.c_void_p(self.torch.cuda.current_stream(self.device).cuda_stream)

    def window_stats(self, beam, window, overlap=0.75):
        """(median, mad) float32 arrays of n_windows + 2 entries; entries 1 .. n_windows hold
        np.median / MAD of the windows [q*shift, min(n, q*shift + window)), the two end entries
        are left 0 (postprocess.bp_threshold_nodes fills them)."""
        t = self.torch
        x = beam.reshape(-1).contiguous()
        n = x.numel()
        window = int(window)
        shift = int((1.0 - overlap) * window)
        nw = self.lib.bpmf_bp_num_windows(n, window, shift)
        if nw == 0:
            raise ValueError("series shorter than the sliding window")
        med = t.zeros(nw + 2, dtype=t.float32, device=self.device)
        mad = t.zeros(nw + 2, dtype=t.float32, device=self.device)
        with t.cuda.device(self.device):
            rc = self.lib.bpmf_bp_window_stats_dev(x.data_ptr(), n, window, shift, self._stream(),
                                                   med.data_ptr(), mad.data_ptr())
        _lib.check(rc, "bpmf_bp_window_stats_dev")
        return med.cpu().numpy(), mad.cpu().numpy()

    def extract_peaks(self, beam, sources, floor, capacity=1 << 18):
        """Records (index, beam, source) of every rising-edge local maximum above `floor`, sorted
        by sample index.  The buffer grows and the extraction is repeated if it overflows."""
        t = self.torch
        x = beam.reshape(-1).contiguous()
        src = sources.reshape(-1).contiguous() if sources is not None else None
        n = x.numel()
        while True:
            count = t.zeros(1, dtype=t.int32, device=self.device)
            rec = t.empty((capacity, 4), dtype=t.int32, device=self.device)
            with t.cuda.device(self.device):
                rc = self.lib.bpmf_bp_extract_peaks_dev(
                    x.data_ptr(), src.data_ptr() if src is not None else None, n, float(floor),
                    capacity, self._stream(), count.data_ptr(), rec.data_ptr())
            _lib.check(rc, "bpmf_bp_extract_peaks_dev")
            found = int(count.item())
            if found <= capacity:
                break
            capacity = found
        out = rec[:found].cpu().numpy().view(peak_dtype).reshape(-1)
        return out[np.argsort(out["index"], kind="stable")]
