"""GPU: the HIP library against the upstream vectors tools/diff_upstream.py --write leaves in
tests/golden/upstream_{mf,bp}.npz (inputs + outputs of the REAL fast_matched_filter / beampower, the
combination of *.compat_* switches under which the CPU oracle is bit-equal to them).  Skipped while the
files do not exist -- the packages are absent from the image this build was made in
(/root/reference/pyproject.toml:28-29 lists them un-vendored)."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


@pytest.mark.parametrize("which", ["mf", "bp"])
def test_library_reproduces_the_upstream_vectors(oracle_lib, hip_opts, which):
    from test_upstream_live import golden_cases
    from seismic_bpmf_amd import beamform, matched_filter
    kit, z, meta, names = golden_cases(which)
    flags_all = meta["equal_on_every_case_flags"]
    for name in names:
        case = {k.split("/", 1)[1]: z[k] for k in z.files if k.startswith(name + "/")}
        case = {k: (v.item() if v.shape == () else v) for k, v in case.items()}
        flags = int(flags_all[0]) if flags_all else (int(case["bit_equal_flags"][0]) if len(case["bit_equal_flags"]) else None)
        if flags is None:
            continue
        for bit, (opt, path) in oracle_lib.COMPAT_OPTIONS.items():
            if path == which:
                hip_opts(opt, 1 if flags & bit else 0)
        if which == "mf":
            got = matched_filter(case["templates"], case["moveouts"], case["weights"], case["data"], int(case["step"]),
                                 arch="gpu", network_sum=bool(case["network_sum"]), check_zeros=False)
            assert kit.compare(case["upstream_out"], got)[0], name
        else:
            got = beamform(case["features"], case["moveouts"], case["weights_phases"], case["weights_sources"],
                           device="gpu", out_of_bounds=str(case["out_of_bounds"]), reduce=str(case["reduce"]))
            up = (case["upstream_beam"], case["upstream_arg"]) if "upstream_beam" in case else case["upstream_out"]
            assert kit.compare(up, got)[0], name
