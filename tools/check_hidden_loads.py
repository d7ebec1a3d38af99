#!/usr/bin/env python
"""Verify the hand-counted row prefetches of k_lanczos3_x2 in the generated gfx950 assembly.

The kernel requests its rows with LDS-DMA loads issued from inline assembly (invisible to the
compiler's s_waitcnt insertion) and waits for them with hand-counted `s_waitcnt vmcnt(N)` statements
tagged `; nus-wait back=K`: "the K-th most recent row request must have landed".  vmcnt retires in
issue order -- after `vmcnt(N)` only the wave's N youngest vector memory instructions may be
outstanding -- so the request has landed exactly when at least N vector memory instructions were
issued after it.  This script checks on the control-flow graph of the generated code what the source
cannot express:

  1. on every path to a tagged wait, at least N vector memory instructions were issued after the
     K-th most recent request (fewer = the row may not be in LDS when it is read: an ERROR);
     exactly N = tight, more than N = the wait also drains younger stores (reported, a time loss);
  2. no other `s_waitcnt vmcnt` sits inside a loop (compiler-placed waits there are what the scheme
     exists to remove), and no vector memory instruction of the loop hides from the count in a
     conditional block (follows from 1 being tight on every path).

Branches on EXEC == 0 only skip regions no lane executes and are not followed; the kernel's requests
and waits sit in wave-uniform control flow.  `s_cbranch_execnz` is followed both ways by default; with
execnz_taken=True it counts as always taken -- for kernels whose loop has no lane-divergent control flow at
all (EXEC never changes there), where the compiler uses it as its "branch always" around the else-side of a
SCALAR branch (k_lanczos3_pq: the two store forms of a row behind `s_cbranch_scc`).

usage: check_hidden_loads.py file.s [kernel-name-substring]   (exit status 0 = ok)
Used by tests/test_kernel_asm.py on a fresh `hipcc -S` of the kernel source.
"""
import re
import sys

VMEM = re.compile(r"^(global_|buffer_|flat_|scratch_)(load|store|atomic)")
CAP = 64   # vmcnt is a 6-bit counter: older than this never matters
KMAX = 16  # deepest request the waits may refer to


def kernel_bodies(asm, want):
    for m in re.finditer(r"^(_Z\w+):[^\n]*\n(.*?)\n\.Lfunc_end", asm, re.S | re.M):
        if want in m.group(1):
            yield m.group(1), m.group(2)


def parse(body):
    """-> list of dicts: kind in dma / wait / cwait / vmem / label / branch / end (other instructions dropped)."""
    ins = []
    in_asm = False
    far = None  # target of a long branch being assembled (s_getpc_b64; s_add_u32 .., (.LBBx_y-.Lpost_getpcN)..; s_setpc_b64)
    for raw in body.split("\n"):
        line = raw.strip()
        if not line:
            continue
        if line.startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if line.startswith(";;#ASMEND"):
            in_asm = False
            continue
        m = re.match(r"^(\.LBB\d+_\d+):", line)
        if m:
            ins.append({"kind": "label", "text": m.group(1)})
            continue
        if line.startswith(";") or line.startswith("."):
            continue
        code, _, comment = line.partition(";")
        code = code.strip()
        if not code:
            continue
        m = re.search(r"\((\.LBB\d+_\d+)-\.Lpost_getpc\d+\)", code)
        if m:
            far = m.group(1)
        if code.startswith("s_setpc_b64") and far:  # a branch beyond the reach of s_branch's 16-bit offset (kernels over 128 KiB)
            ins.append({"kind": "branch", "text": "s_branch " + far, "target": far, "cond": False, "never": False})
            far = None
            continue
        if code.startswith("global_load_lds") or (code.startswith("buffer_load") and code.endswith(" lds")):
            ins.append({"kind": "dma", "text": code, "hidden": in_asm})
        elif code.startswith("s_waitcnt") and "vmcnt" in code:
            n = int(re.search(r"vmcnt\((\d+)\)", code).group(1))
            tag = re.search(r"nus-wait back=(\d+)", comment)
            if in_asm and tag:
                ins.append({"kind": "wait", "text": code + " ;" + comment, "n": n, "back": int(tag.group(1))})
            else:
                ins.append({"kind": "cwait", "text": code, "n": n})
        elif VMEM.match(code):
            ins.append({"kind": "vmem", "text": code})
        elif code.startswith("s_cbranch") or code.startswith("s_branch"):
            ins.append({"kind": "branch", "text": code, "target": code.split()[-1], "cond": code.startswith("s_cbranch"),
                        "never": code.startswith("s_cbranch_execz")})
        elif code.startswith("s_endpgm"):
            ins.append({"kind": "end", "text": code})
    return ins


def blocks_of(ins, execnz_taken=False):
    """Basic blocks as (start, end_exclusive) and successor lists."""
    starts = {0}
    for i, x in enumerate(ins):
        if x["kind"] == "label":
            starts.add(i)
        if x["kind"] in ("branch", "end") and i + 1 < len(ins):
            starts.add(i + 1)
    starts = sorted(starts)
    label_at = {x["text"]: i for i, x in enumerate(ins) if x["kind"] == "label"}
    blocks = [(s, starts[k + 1] if k + 1 < len(starts) else len(ins)) for k, s in enumerate(starts)]
    index_of = {s: k for k, (s, _) in enumerate(blocks)}
    succ = []
    for (s, e) in blocks:
        last = ins[e - 1]
        out = []
        if last["kind"] == "branch":
            if not last["never"]:
                out.append(index_of[label_at[last["target"]]])
            always = execnz_taken and last["text"].startswith("s_cbranch_execnz")
            if last["cond"] and e < len(ins) and not always:
                out.append(index_of[e])
        elif last["kind"] != "end" and e < len(ins):
            out.append(index_of[e])
        succ.append(out)
    return blocks, succ


def check(body, cap=CAP, kmax=KMAX, execnz_taken=False):
    """cap / kmax: ages are counted up to `cap` and the `kmax` most recent requests are tracked.  The defaults suit kernels whose
    loop issues the same instructions on every path (k_lanczos3_x2: the tightness of every wait is checked too); a kernel with
    stores on some paths only (k_resize_down: its waits are deliberately conservative) needs just "never fewer than N" and gets a
    small cap so that the path histories stay few."""
    CAP, KMAX = cap, kmax  # noqa: N806 (shadow the module defaults below)
    ins = parse(body)
    if not ins:
        return {"errors": ["empty kernel body"], "requests": 0, "hand_waits": 0, "compiler_vmcnt_waits_in_loops": [],
                "waits_not_tight": {}}
    blocks, succ = blocks_of(ins, execnz_taken)
    errors, seen = set(), {}
    fresh = (CAP,) * KMAX  # ages[j] = vector memory instructions issued after the (j+1)-th most recent request
    state_in = [None] * len(blocks)  # set of age vectors (one per distinct path history)
    state_in[0] = frozenset([fresh])
    work = [0]
    while work:
        b = work.pop()
        st = set(state_in[b])
        for i in range(*blocks[b]):
            x = ins[i]
            k = x["kind"]
            if k == "dma":
                st = {(0,) + tuple(min(a + 1, CAP) for a in ages[:-1]) for ages in st}
            elif k == "vmem":
                st = {tuple(min(a + 1, CAP) for a in ages) for ages in st}
            elif k == "wait":
                n, back = x["n"], x["back"]
                if n == 0:
                    st = {fresh}
                else:
                    for ages in st:
                        seen.setdefault(i, set()).add(ages[back - 1])
                        if ages[back - 1] < n:
                            errors.add(f"`{x['text'].strip()}` (#{i}): on some path only {ages[back - 1]} vector memory "
                                       f"instructions follow the request it waits for: vmcnt({n}) does not retire it")
            elif k == "cwait":
                if x["n"] == 0:
                    st = {fresh}
            elif k == "end":
                st = set()
        st = frozenset(st)
        for t in succ[b]:
            new = st if state_in[t] is None else (state_in[t] | st)
            if new != state_in[t]:
                if len(new) > 4096:
                    errors.add("path explosion: the loop's memory instructions differ from path to path")
                    continue
                state_in[t] = new
                work.append(t)
    # compiler-placed vmcnt waits inside a loop (a block that can reach itself)
    loop_blocks = set()
    for b in range(len(blocks)):
        visited, stack = set(), list(succ[b])
        while stack:
            t = stack.pop()
            if t == b:
                loop_blocks.add(b)
                break
            if t not in visited:
                visited.add(t)
                stack.extend(succ[t])
    cwaits = [x["text"] for b in sorted(loop_blocks) for x in ins[blocks[b][0]:blocks[b][1]] if x["kind"] == "cwait"]
    # (CAP = the request was already retired by an earlier full drain: the first steps after the loop's entry)
    loose = {ins[i]["text"].strip() + f" (#{i})": sorted(cs) for i, cs in seen.items()
             if any(c != ins[i]["n"] and c != CAP for c in cs)}
    return {"errors": sorted(errors), "requests": sum(x["kind"] == "dma" and x["hidden"] for x in ins),
            "hand_waits": sum(x["kind"] == "wait" for x in ins), "compiler_vmcnt_waits_in_loops": cwaits,
            "waits_not_tight": loose}


def main():
    path = sys.argv[1]
    want = sys.argv[2] if len(sys.argv) > 2 else "k_lanczos3_x2IL"
    asm = open(path).read()
    bad = found = 0
    for name, body in kernel_bodies(asm, want):
        found += 1
        r = check(body, execnz_taken="k_lanczos3_pq" in name)
        print(f"{name}: {r['requests']} hidden row requests, {r['hand_waits']} hand-counted waits, "
              f"{len(r['compiler_vmcnt_waits_in_loops'])} compiler vmcnt waits in loops, "
              f"{len(r['waits_not_tight'])} waits not tight, {len(r['errors'])} error(s)")
        for e in r["errors"][:20]:
            print("  ERROR " + e)
        for t, cs in list(r["waits_not_tight"].items())[:12]:
            print(f"  note  {t}: instructions issued since the request, by path: {cs}")
        for t in r["compiler_vmcnt_waits_in_loops"][:8]:
            print(f"  note  compiler-placed in a loop: {t}")
        bad += len(r["errors"])
    if not found:
        print("no kernel matched", want)
        return 2
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
