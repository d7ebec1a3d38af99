#!/usr/bin/env python
"""Experiment (round 3): swap the two multiplicand VGPRs of commutative VOP2 instructions (v_fmac_f32 / v_mul_f32 / v_add_f32,
e32 encodings, plain VGPR operands) in a device assembly file by a register-bank rule, to see whether VGPR bank conflicts
between an instruction's first source and its destination cost anything in the real kernels.  The arithmetic is unchanged
bit for bit (a*b = b*a).   usage: asm_swap.py {avoid|seek} in.s out.s
  avoid: make bank(src0) != bank(dst) where a swap achieves it;  seek: make bank(src0) == bank(dst) where a swap achieves it."""
import re
import sys

PAT = re.compile(r"^(\s*)(v_fmac_f32_e32|v_mul_f32_e32|v_add_f32_e32)(\s+)v(\d+),\s*v(\d+),\s*v(\d+)(\s*(;.*)?)$")


def main():
    rule, src, dst = sys.argv[1], sys.argv[2], sys.argv[3]
    n = swapped = 0
    out = []
    for line in open(src):
        m = PAT.match(line.rstrip("\n"))
        if m:
            n += 1
            d, a, b = int(m.group(4)), int(m.group(5)), int(m.group(6))
            bad_now = (a % 4 == d % 4)
            bad_swapped = (b % 4 == d % 4)
            do = (rule == "avoid" and bad_now and not bad_swapped) or (rule == "seek" and not bad_now and bad_swapped)
            if do:
                swapped += 1
                line = f"{m.group(1)}{m.group(2)}{m.group(3)}v{d}, v{b}, v{a}{m.group(7) or ''}\n"
        out.append(line)
    open(dst, "w").writelines(out)
    print(f"{src}: {n} commutative VOP2 instructions, {swapped} swapped ({rule})")


if __name__ == "__main__":
    main()
