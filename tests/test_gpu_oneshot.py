"""AGX_FLAG_ONE_SHOT (-m gpu): the application's flow — a unit is uploaded once and its download lands in the pinned memory of its dead
inputs — gives the bytes of the oracle, from text and from the unit cache, and refuses a second upload until the inputs are handed over again."""
import os

import pytest

import harness as H

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def agx():
    import aligngraph_amd as A
    if not os.path.exists(A.LIB_PATH):
        from aligngraph_amd import build as B
        B.build()
    assert A.device_count() > 0, "no HIP device: the gpu tests must run on the MI355X box"
    return A


def check(agx, tmp, want, cov, expect_cache):
    with agx.Unit(k=5, insert_variation=50, coverage=cov, flags=agx.AGX_FLAG_ONE_SHOT) as u:
        u.load_files(tmp, 0)
        assert u.stats()["from_cache"] == expect_cache
        u.upload(); u.build()
        got = u.finish()
        for key in ("initial", "pre", "extended"):
            assert got[key] == want[key], key
        assert u.finish() == got                 # (downloads again — the walk consumed the first download — into the same borrowed memory)
        with pytest.raises(agx.AgxError) as e:
            u.upload()
        assert e.value.code == agx.AGX_E_ARG and "one-shot" in e.value.msg
        # nor may staging (or the size query, which stages what is not staged) clear the way for one: the fast loader's staged arrays ARE the buffers the
        # download landed in, and a unit out of its cache file has nothing left to stage from
        for again in (u.stage, u.hbm_needed):
            with pytest.raises(agx.AgxError) as e:
                again()
            assert e.value.code == agx.AGX_E_ARG
        with pytest.raises(agx.AgxError) as e:
            u.upload()
        assert e.value.code == agx.AGX_E_ARG and "one-shot" in e.value.msg
        u.load_files(tmp, 0)                     # the inputs handed over again: a new unit
        u.upload(); u.build()
        assert u.finish() == got


@pytest.mark.parametrize("pairs", [40000, 1500], ids=["deep", "thin"])      # thin: the staged inputs are smaller than the walk graph, most of the download gets buffers of its own
def test_one_shot_unit_gives_the_oracle_bytes(agx, built, tmp_path, monkeypatch, pairs):
    run = H.synth(str(tmp_path / "run"), seed=300 + pairs % 7, chroms="200000", pairs=pairs, coverage=3, read_indel=0.2, multi=0.2, sam_seq=0)
    tmp = os.path.join(run, "tmp")
    want = H.run_oracle(tmp, 0, 5, 50, 3)
    monkeypatch.setenv("AGX_NO_CACHE", "1")
    check(agx, tmp, want, 3, 0)
    monkeypatch.delenv("AGX_NO_CACHE")
    agx.cache_build(tmp, 0)
    check(agx, tmp, want, 3, 1)
