"""the drop-in claim of INTEGRATION.md §3 exercised against the REAL reference classes.
CPU-only, no kernel launch; runs where /root/reference exists (the build container) and is skipped elsewhere."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REFERENCE = os.environ.get("SIAMMOT_REFERENCE", "/root/reference")


@pytest.mark.skipif(not os.path.isdir(os.path.join(REFERENCE, "siammot")), reason="reference checkout not present")
def test_reference_build_track_head_holds_the_hip_emm():
    """A fresh interpreter (the stubs must not leak into this process): reference ``build_track_head`` ->
    reference ``TrackHead`` holding ``siammot_amd.emm.EMM``; reference state_dict loads; reference TrackHead /
    TrackSolver / TrackPool drive it through ``extract_cache`` / ``forward`` with the reference's call structure."""
    res = subprocess.run([sys.executable, os.path.join(ROOT, "oracle", "check_reference_dropin.py")],
                         capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert res.returncode == 0, res.stderr[-3000:]
    d = json.loads([ln for ln in res.stdout.splitlines() if ln.startswith("{")][-1])
    assert d["ok"] and d["tracker_class"] == "siammot_amd.emm.EMM"
    assert d["track_head_class"] == "siammot.modelling.track_head.track_head.TrackHead"
    assert d["state_dict_keys"] == 12 and d["tracks_started"] == [0, 1, -1]
