"""The hand-counted row prefetches of k_lanczos3_x2 (nus_k_lanczos_x2.hip), checked on the code hipcc
generates: every `s_waitcnt vmcnt(N)` the kernel places by hand must retire the row request it is for on
every path through the unrolled loop, be tight (not drain younger stores), and be the only vmcnt wait inside
the loop.  tools/check_hidden_loads.py does the control-flow analysis; this test compiles the kernel source
to assembly (no GPU needed) and runs it for all ten instantiations (EXACT x BLEND, plus the one-launch UNIT forms of the blend kernels)."""
import os
import subprocess
import sys

import pytest

from conftest import ROOT

sys.path.insert(0, os.path.join(ROOT, "tools"))

CSRC = os.path.join(ROOT, "nu_scaler_amd", "csrc")


@pytest.fixture(scope="module")
def kernel_asm(tmp_path_factory):
    out = tmp_path_factory.mktemp("asm") / "nus_k_lanczos_x2.s"
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-x", "hip",
           "--cuda-device-only", "-S", "-I", CSRC, "-I", os.path.join(ROOT, "include"), "-o", str(out),
           os.path.join(CSRC, "nus_k_lanczos_x2.hip")]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr
    return out.read_text()


def test_flags_match_the_makefile():
    """The assembly checked here is only meaningful if it is built like the library."""
    mk = open(os.path.join(CSRC, "Makefile")).read()
    assert "-O3 -std=c++17 -fPIC -ffp-contract=off" in mk


def test_hand_counted_waits_of_every_instantiation(kernel_asm):
    import check_hidden_loads as chk

    bodies = list(chk.kernel_bodies(kernel_asm, "k_lanczos3_x2IL"))
    assert len(bodies) == 10, [n for n, _ in bodies]  # EXACT x BLEND, + EXACT x BLEND {1, 2} x UNIT
    for name, body in bodies:
        r = chk.check(body)
        blend = "ELi0E" not in name
        assert r["errors"] == [], (name, r["errors"][:3])
        assert r["compiler_vmcnt_waits_in_loops"] == [], (name, r["compiler_vmcnt_waits_in_loops"][:3])
        assert r["waits_not_tight"] == {}, (name, r["waits_not_tight"])
        # 6 unrolled steps + the first requests of a block; one hand wait per step + the drain at loop entry
        assert r["requests"] == (16 if blend else 8), (name, r["requests"])
        assert r["hand_waits"] == 7, (name, r["hand_waits"])


def test_checker_catches_a_wait_that_is_too_loose():
    import check_hidden_loads as chk

    body = """
	s_mov_b32 m0, s4
	;;#ASMSTART
	global_load_lds_dwordx4 v1, s[0:1]
	;;#ASMEND
	buffer_store_dwordx4 v[0:3], v9, s[12:15], 0 offen
	;;#ASMSTART
	s_waitcnt vmcnt(2) ; nus-wait back=1
	;;#ASMEND
	ds_read_b128 v[4:7], v2
	s_endpgm
"""
    r = chk.check(body)
    assert len(r["errors"]) == 1 and "does not retire" in r["errors"][0]
    ok = chk.check(body.replace("vmcnt(2)", "vmcnt(1)"))
    assert ok["errors"] == [] and ok["waits_not_tight"] == {}
    strict = chk.check(body.replace("vmcnt(2)", "vmcnt(0) ; x").replace("nus-wait back=1", "nus-wait back=1"))
    assert strict["errors"] == []


def test_checker_sees_a_store_hidden_in_a_conditional_block():
    import check_hidden_loads as chk

    body = """
	;;#ASMSTART
	global_load_lds_dwordx4 v1, s[0:1]
	;;#ASMEND
	s_cbranch_scc1 .LBB0_2
	buffer_store_dwordx4 v[0:3], v9, s[12:15], 0 offen
.LBB0_2:
	buffer_store_dwordx4 v[0:3], v9, s[12:15], 0 offen
	;;#ASMSTART
	s_waitcnt vmcnt(2) ; nus-wait back=1
	;;#ASMEND
	s_endpgm
"""
    r = chk.check(body)
    assert r["errors"], "one path issues a single store: vmcnt(2) does not retire the request there"
