/* C caller doing real work through the boundary (run on the GPU box by tests/test_gpu_parity.py):
 * nearest x2 must replicate pixels, the zero-flow in-between frame of 255 and 0 must be 127
 * (the truncation the reference's interp_half.png pins). */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "nuscaler_hip.h"

#define CHECK(cond)                                                  \
    do {                                                             \
        if (!(cond)) {                                               \
            fprintf(stderr, "abi_upscale: %s failed (line %d): %s\n", #cond, __LINE__, nus_last_error()); \
            return 1;                                                \
        }                                                            \
    } while (0)

int main(void)
{
    enum { W = 16, H = 8 };
    unsigned char in[W * H * 4], out[2 * W * 2 * H * 4];
    int x, y, c;
    for (y = 0; y < H; ++y)
        for (x = 0; x < W; ++x)
            for (c = 0; c < 4; ++c) in[(y * W + x) * 4 + c] = (unsigned char)(x * 13 + y * 7 + c * 31);

    nus_upscaler *u = nus_upscaler_create(NUS_ALG_NEAREST, NUS_QUALITY_QUALITY);
    CHECK(u != NULL);
    CHECK(nus_upscaler_initialize(u, W, H, 2 * W, 2 * H) == NUS_OK);
    CHECK(nus_upscaler_output_size(u) == sizeof out);
    CHECK(nus_upscaler_upscale(u, in, sizeof in, out, sizeof out) == NUS_OK);
    for (y = 0; y < 2 * H; ++y)
        for (x = 0; x < 2 * W; ++x)
            CHECK(memcmp(&out[(y * 2 * W + x) * 4], &in[((y / 2) * W + x / 2) * 4], 4) == 0);
    CHECK(nus_upscaler_upscale(u, in, sizeof in - 4, out, sizeof out) == NUS_ERR_SIZE_MISMATCH);
    nus_upscaler_destroy(u);

    {
        unsigned char a[W * H * 4], b[W * H * 4], mid[W * H * 4];
        size_t i;
        memset(a, 255, sizeof a);
        memset(b, 0, sizeof b);
        nus_interp *it = nus_interp_create(NUS_WG_WIDE_32X8);
        CHECK(it != NULL);
        CHECK(nus_interp_interpolate(it, a, sizeof a, b, sizeof b, NULL, W, H, 0.5f, mid, sizeof mid) == NUS_OK);
        for (i = 0; i < sizeof mid; ++i) CHECK(mid[i] == 127);
        nus_interp_destroy(it);
    }
    printf("abi_upscale ok\n");
    return 0;
}
