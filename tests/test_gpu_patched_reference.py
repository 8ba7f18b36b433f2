"""The drop-in claim at byte level (-m gpu): the REFERENCE ITSELF with the patch of INTEGRATION.md §1 applied — its five calls per unit (AG:4768-4776)
replaced by one agx_run_unit — built in the build container into oracle/_ref/AlignGraph_patched (oracle/harness.py build_patched_reference; a git-ignored
artefact that rides to the GPU box like the unpatched reference binaries), run here through `--resume` on a two-unit job beside the unpatched reference:
the three per-unit files of every unit and the run's final output must be the same bytes."""
import os

import pytest

import harness as H

pytestmark = pytest.mark.gpu


@pytest.mark.skipif(not (os.path.exists(H.REF_PATCHED) and os.path.exists(H.REF_O2)), reason="oracle/_ref holds no patched reference (built where /root/reference exists)")
def test_the_patched_reference_produces_the_reference_bytes(built, tmp_path):
    import aligngraph_amd as A
    assert A.device_count() > 0
    run = H.synth(str(tmp_path / "run"), seed=41, chroms="60000,35000", pairs=30000, coverage=4, contig_min=1200, contig_max=4000, read_indel=0.2, read_clip=0.1, multi=0.1, sam_seq=1)
    want, _ = H.run_reference(run, opt=True, keep=True)
    got, _ = H.run_reference(run, exe=H.REF_PATCHED, suffix=".patched", keep=True)
    assert len(want) == len(got) == 2
    for u in range(2):
        for key in ("initial", "pre", "extended"):
            assert got[u][key] == want[u][key], "unit %d: %s differs between the reference and the reference patched to call agx_run_unit" % (u, key)
        assert want[u]["extended"].count(b">") > 0
    for name in ("extended.fa", "remaining.fa"):                      # what refinement (AG:4785) makes of them: the run's final output
        a, b = os.path.join(run + ".ref", name), os.path.join(run + ".patched", name)
        assert os.path.exists(a) and open(a, "rb").read() == open(b, "rb").read(), name
