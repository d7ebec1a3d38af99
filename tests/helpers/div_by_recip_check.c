/* Exhaustive check of the kernels' division shortcut (nus_device.hpp: div_by_recip): q = RN(x z), r = fma(-y, q, x),
 * result = fma(r, z, q) with z = RN(1 / y) against the IEEE quotient x / y, over every mantissa of x (three binades, both
 * signs) for the divisors the kernels use it with: 9 (Horn-Schunck mean), 255 (unorm8), the all-ones mantissa (the
 * exception of Markstein's theorem, which a Horn-Schunck denominator can hit) and a spread of ordinary mantissas.
 * The sequence is invariant under scaling x or y by a power of two (normal range), so mantissas are all that matters.
 * Prints "ok <checked>" or "bad <count> ..."; test infrastructure only. */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

static float asf(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
static uint32_t asu(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static float div_by_recip(float x, float y, float z)
{
    const float q = x * z;
    const float r = fmaf(-y, q, x);
    return fmaf(r, z, q);
}

int main(void)
{
    uint32_t ys[40];
    int ny = 0;
    ys[ny++] = asu(9.0f);
    ys[ny++] = asu(255.0f);
    ys[ny++] = 0x3FFFFFFFu; /* mantissa all ones */
    ys[ny++] = 0x3FFFFFFEu;
    ys[ny++] = 0x3F800001u;
    uint32_t s = 12345u;
    while (ny < 40) { s = s * 1664525u + 1013904223u; ys[ny++] = 0x3F800000u | (s >> 9); }
    unsigned long long checked = 0, bad = 0;
    for (int k = 0; k < ny; ++k) {
        const float y = asf(ys[k]), z = 1.0f / y;
        const uint32_t step = k < 3 ? 1u : 7u; /* the three named divisors exhaustively, the others every 7th mantissa */
        for (uint32_t e = 0; e < 3; ++e)
            for (uint32_t m = 0; m < (1u << 23); m += step) {
                const float x = asf(((126u + e) << 23) | m);
                if (asu(div_by_recip(x, y, z)) != asu(x / y)) ++bad;
                if (asu(div_by_recip(-x, y, z)) != asu(-x / y)) ++bad;
                checked += 2;
            }
    }
    if (bad) printf("bad %llu of %llu\n", bad, checked);
    else printf("ok %llu\n", checked);
    return bad != 0;
}
