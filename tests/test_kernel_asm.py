"""The hand-counted row prefetches of k_lanczos3_x2 (nus_k_lanczos_x2.hip), checked on the code hipcc
generates: every `s_waitcnt vmcnt(N)` the kernel places by hand must retire the row request it is for on
every path through the unrolled loop, be tight (not drain younger stores), and be the only vmcnt wait inside
the loop.  tools/check_hidden_loads.py does the control-flow analysis; this test compiles the kernel source
to assembly (no GPU needed) and runs it for all twelve instantiations (EXACT x BLEND, plus the one-launch UNIT forms of the blend kernels)."""
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
    assert len(bodies) == 12, [n for n, _ in bodies]  # EXACT x BLEND, + EXACT x BLEND {1, 2} x UNIT, + EXACT x NARROW (4 taps)
    for name, body in bodies:
        r = chk.check(body)
        blend = "ELi0E" not in name
        assert r["errors"] == [], (name, r["errors"][:3])
        assert r["compiler_vmcnt_waits_in_loops"] == [], (name, r["compiler_vmcnt_waits_in_loops"][:3])
        assert r["waits_not_tight"] == {}, (name, r["waits_not_tight"])
        # 6 unrolled steps + the first requests of a block; one hand wait per step + the drain at loop entry
        assert r["requests"] == (16 if blend else 8), (name, r["requests"])
        assert r["hand_waits"] == 7, (name, r["hand_waits"])


def test_lanczos_r32_row_ring_waits():
    """k_lanczos3_r32 (x3/2: 720p -> 1080p, 1440p -> 4K) carries the same scheme: two row requests and six stores per step of two
    input rows, one hand-counted wait per request, tight on every path; nothing else waits on vmcnt inside its loop (the row
    pair's weight class is read with a scalar load for that reason)."""
    import check_hidden_loads as chk

    out = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-x", "hip",
                          "--cuda-device-only", "-S", "-I", CSRC, "-I", os.path.join(ROOT, "include"), "-o", "-",
                          os.path.join(CSRC, "nus_k_lanczos_r32.hip")], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr
    bodies = list(chk.kernel_bodies(out.stdout, "k_lanczos3_r32IL"))
    assert len(bodies) == 2, [n for n, _ in bodies]  # EXACT, FMA
    for name, body in bodies:
        r = chk.check(body)
        assert r["errors"] == [], (name, r["errors"][:3])
        assert r["compiler_vmcnt_waits_in_loops"] == [], (name, r["compiler_vmcnt_waits_in_loops"][:3])
        assert r["waits_not_tight"] == {}, (name, r["waits_not_tight"])
        # 3 unrolled steps x 2 requests + the first 2 D of a block; one hand wait per request + the drain at loop entry
        assert r["requests"] == 10 and r["hand_waits"] == 7, (name, r["requests"], r["hand_waits"])


def test_lanczos_r43_row_ring_waits():
    """k_lanczos3_r43 (x4/3: 1080p -> 1440p): three 12-byte row requests (global_load_lds_dwordx3) and four stores per step,
    one hand-counted wait per request, tight on every path."""
    import check_hidden_loads as chk

    out = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-x", "hip",
                          "--cuda-device-only", "-S", "-I", CSRC, "-I", os.path.join(ROOT, "include"), "-o", "-",
                          os.path.join(CSRC, "nus_k_lanczos_r43.hip")], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr
    bodies = list(chk.kernel_bodies(out.stdout, "k_lanczos3_r43IL"))
    assert len(bodies) == 2, [n for n, _ in bodies]  # EXACT, FMA
    for name, body in bodies:
        r = chk.check(body)
        assert r["errors"] == [], (name, r["errors"][:3])
        assert r["compiler_vmcnt_waits_in_loops"] == [], (name, r["compiler_vmcnt_waits_in_loops"][:3])
        assert r["waits_not_tight"] == {}, (name, r["waits_not_tight"])
        # 2 unrolled steps x 3 requests + the first 3 D of a block; one hand wait per request + the drain at loop entry
        assert r["requests"] == 9 and r["hand_waits"] == 7, (name, r["requests"], r["hand_waits"])
        assert "global_load_lds_dwordx3" in body


def test_lanczos_pq_row_ring_waits():
    """k_lanczos3_pq (x5/4, x6/5, x7/5, x8/5, x9/5, x5/3, x5/2, x7/2): Q row requests of 1 - 2 LDS-DMA pieces and P rows of 2 - 3 stores per step,
    one hand-counted wait per row request, tight on every path of every instantiation.  (The kernel's first form also had a second,
    direct way of storing a row behind a scalar branch: there the checker caught the compiler merging an 8- and a 4-byte store into
    one 12-byte instruction at P = 7, and later a branch pair turned into a flag that no control-flow analysis can follow -- the
    reason there is one store form now.)"""
    import check_hidden_loads as chk

    from concurrent.futures import ThreadPoolExecutor

    def asm_of(unit):  # (the Q = 5 factors are translation units of their own: a minute of compile time each)
        out = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-x", "hip",
                              "--cuda-device-only", "-S", "-I", CSRC, "-I", os.path.join(ROOT, "include"), "-o", "-",
                              os.path.join(CSRC, unit)], capture_output=True, text=True, timeout=1500)
        assert out.returncode == 0, out.stderr
        return out.stdout

    units = ["nus_k_lanczos_pq.hip", "nus_k_lanczos_pq_65.hip", "nus_k_lanczos_pq_75.hip", "nus_k_lanczos_pq_85.hip", "nus_k_lanczos_pq_95.hip"]
    with ThreadPoolExecutor(5) as pool:
        bodies = [b for text in pool.map(asm_of, units) for b in chk.kernel_bodies(text, "k_lanczos3_pqIL")]
    assert len(bodies) == 32, [n for n, _ in bodies]  # (EXACT, FMA) x (6-tap, 4-tap: round 6) x eight factors
    # (P, Q) -> unrolled steps, LDS-DMA pieces per row request
    shape = {(5, 4): (3, 1), (6, 5): (6, 2), (5, 3): (2, 1), (5, 2): (3, 2), (7, 2): (3, 2), (7, 5): (6, 2), (8, 5): (6, 2), (9, 5): (6, 2)}
    for name, body in bodies:
        import re

        m = re.search(r"ILb[01]ELi(\d)ELi(\d)E", name)
        P, Q = int(m.group(1)), int(m.group(2))
        steps, pieces = shape[(P, Q)]
        # the loop has no lane-divergent control flow (its stores are range-checked, its branches scalar)
        loop = body[body.index("nus-wait back=0"):]  # everything behind the drain at the loop's entry
        assert "s_cbranch_scc" in loop and "v_cmpx" not in loop and "saveexec" not in loop, name
        r = chk.check(body, execnz_taken=True)
        assert r["errors"] == [], (name, r["errors"][:3])
        assert r["compiler_vmcnt_waits_in_loops"] == [], (name, r["compiler_vmcnt_waits_in_loops"][:3])
        assert r["waits_not_tight"] == {}, (name, r["waits_not_tight"])
        # Q requests per unrolled step + the first Q of a block; one hand wait per row request + the drain at loop entry
        assert r["requests"] == (steps + 1) * Q * pieces and r["hand_waits"] == steps * Q + 1, (name, r["requests"], r["hand_waits"])


def test_lanczos_xs_row_ring_waits():
    """k_lanczos3_xs (x3, x4: 720p / 540p -> 4K): one row request and S x S stores per step, one hand-counted wait per step,
    tight on every path of all six instantiations, no other vmcnt wait inside the loop."""
    import check_hidden_loads as chk

    out = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-x", "hip",
                          "--cuda-device-only", "-S", "-I", CSRC, "-I", os.path.join(ROOT, "include"), "-o", "-",
                          os.path.join(CSRC, "nus_k_lanczos_xs.hip")], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr
    bodies = list(chk.kernel_bodies(out.stdout, "k_lanczos3_xsIL"))
    assert len(bodies) == 6, [n for n, _ in bodies]  # EXACT x {x4, x3 with weight classes, x3 without}
    for name, body in bodies:
        r = chk.check(body)
        assert r["errors"] == [], (name, r["errors"][:3])
        assert r["compiler_vmcnt_waits_in_loops"] == [], (name, r["compiler_vmcnt_waits_in_loops"][:3])
        assert r["waits_not_tight"] == {}, (name, r["waits_not_tight"])
        assert r["requests"] == 3 and r["hand_waits"] == 2, (name, r["requests"], r["hand_waits"])  # first D = 2 + 1 per step; drain + 1


def test_resize_win_row_ring_waits():
    """k_resize_win (every up-scaling factor without a kernel of its own): rows through an LDS-DMA ring, requested two window
    advances ahead; the one hand-placed wait relies on at least D stores between a request and its use (an advance happens at most
    once per output row): never fewer on any path, no compiler-placed vmcnt wait left in the loop.  The shape whose union weights
    fill the LDS (3 columns per lane + union H pass) keeps ordinary loads and has no hidden requests."""
    import re

    import check_hidden_loads as chk

    out = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-x", "hip",
                          "--cuda-device-only", "-S", "-I", CSRC, "-I", os.path.join(ROOT, "include"), "-o", "-",
                          os.path.join(CSRC, "nus_k_resize.hip")], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr
    bodies = list(chk.kernel_bodies(out.stdout, "k_resize_winIL"))
    assert len(bodies) == 20, len(bodies)  # EXACT x VC {2, 3} x (UNION {0, 10} x N {4, 2} + UNION 8 at N = 2)
    ringed = 0
    for name, body in bodies:
        vc, uni = (int(v) for v in re.search(r"k_resize_winILb[01]ELi(\d)ELi(\d+)E", name).groups())
        r = chk.check(body, cap=16, kmax=8)
        assert r["errors"] == [], (name, r["errors"][:3])
        if vc == 3 and uni:
            assert r["requests"] == 0 and r["hand_waits"] == 0, name
            continue
        ringed += 1
        # first 2 rows + 1 per advance; round 5: the walk is written out for the window's seven positions (it rotates instead of
        # shifting: the slot indices must be compile-time constants), each with its own advance
        assert r["requests"] == (2 + 7) * vc and r["hand_waits"] == 1 + 7, (name, r["requests"], r["hand_waits"])
        assert r["compiler_vmcnt_waits_in_loops"] == [], (name, r["compiler_vmcnt_waits_in_loops"][:3])
        for t, counts in r["waits_not_tight"].items():
            n = int(re.search(r"vmcnt\((\d+)\)", t).group(1))
            assert min(counts) >= n, (name, t, counts)
        assert re.search(r"s_waitcnt lgkmcnt\(0\)[^\n]*\n(?:[^\n]*\n){0,16}?[^\n]*global_load_lds_dword", body), name
    assert ringed == 14


def test_resize_down_row_ring_waits():
    """k_resize_down's LDS-DMA row ring (footprints of up to two columns per lane): its one hand-placed wait per row must retire
    the row it is about to read on every path -- stores of completed output rows sit between the requests on some paths only, so
    the wait is conservative rather than tight -- the slot's reads are waited for before the slot is requested again, and the
    instantiations without the ring have no hidden requests at all."""
    import re

    import check_hidden_loads as chk

    out = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-x", "hip",
                          "--cuda-device-only", "-S", "-I", CSRC, "-I", os.path.join(ROOT, "include"), "-o", "-",
                          os.path.join(CSRC, "nus_k_resize_down.hip")], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr
    bodies = list(chk.kernel_bodies(out.stdout, "k_resize_downIL"))
    assert len(bodies) == 20, len(bodies)  # EXACT x VC 1..5 x HT {16, 32}
    ringed = 0
    for name, body in bodies:
        vc = int(re.search(r"k_resize_downILb[01]ELi(\d)E", name).group(1))
        r = chk.check(body, cap=12, kmax=8)
        assert r["errors"] == [], (name, r["errors"][:3])
        if vc <= 2:
            ringed += 1
            assert r["requests"] == 5 * vc and r["hand_waits"] == 1, (name, r["requests"], r["hand_waits"])  # 4 rows ahead + 1 per step
            for t, counts in r["waits_not_tight"].items():  # every path has AT LEAST the N instructions the wait relies on
                n = int(re.search(r"vmcnt\((\d+)\)", t).group(1))
                assert min(counts) >= n, (name, t, counts)
            # the reads of a slot return before the slot is requested again
            assert re.search(r"ds_read\w*_b32[^\n]*\n(?:[^\n]*\n){0,12}?[^\n]*s_waitcnt lgkmcnt\(0\)[^\n]*\n(?:[^\n]*\n){0,6}?[^\n]*global_load_lds_dword", body), name
        else:
            assert r["requests"] == 0 and r["hand_waits"] == 0, name
    assert ringed == 8


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
