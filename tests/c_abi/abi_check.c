/* Compiled as plain C (gcc -std=c99 -Wall -Wextra -Werror -pedantic) against include/nuscaler_hip.h and linked
 * with libnuscaler_hip.so: the boundary must be usable from C, not only through ctypes.  No compute calls
 * (this runs on machines without a GPU); the handles are created, configured, queried and destroyed. */
#include <stdio.h>
#include <string.h>

#include "nuscaler_hip.h"

#define CHECK(cond)                                                  \
    do {                                                             \
        if (!(cond)) {                                               \
            fprintf(stderr, "abi_check: %s failed (line %d)\n", #cond, __LINE__); \
            return 1;                                                \
        }                                                            \
    } while (0)

int main(void)
{
    CHECK(nus_abi_version() == 1);
    CHECK(strcmp(nus_status_string(NUS_ERR_NO_DEVICE), "no HIP device") == 0);

    nus_upscaler *u = nus_upscaler_create(NUS_ALG_LANCZOS3, NUS_QUALITY_QUALITY);
    CHECK(u != NULL);
    CHECK(strcmp(nus_upscaler_name(u), "HipLanczos3Upscaler") == 0);
    CHECK(nus_upscaler_algorithm(u) == NUS_ALG_LANCZOS3);
    CHECK(nus_upscaler_set_quality(u, NUS_QUALITY_BALANCED) == NUS_OK);
    CHECK(nus_upscaler_quality(u) == NUS_QUALITY_BALANCED);
    CHECK(nus_upscaler_set_input_format(u, NUS_FORMAT_BGRX8) == NUS_OK);
    CHECK(nus_upscaler_set_input_format(u, 17) == NUS_ERR_INVALID_ARGUMENT);
    CHECK(nus_upscaler_set_option(u, "force_general", 1) == NUS_OK);
    CHECK(nus_upscaler_set_option(u, "no_such_option", 1) == NUS_ERR_INVALID_ARGUMENT);
    CHECK(strstr(nus_upscaler_last_error(u), "no_such_option") != NULL);
    {
        unsigned char in[16] = {0}, out[64];
        /* not initialized: the reference's text (upscale/mod.rs:937-939) */
        CHECK(nus_upscaler_upscale(u, in, sizeof in, out, sizeof out) == NUS_ERR_NOT_INITIALIZED);
        CHECK(strcmp(nus_upscaler_last_error(u), "Upscaler not initialized. Call initialize() first.") == 0);
        CHECK(nus_upscaler_output_size(u) == 0);
    }
    nus_upscaler_destroy(u);
    CHECK(nus_upscaler_create(99, NUS_QUALITY_QUALITY) == NULL);

    nus_upscaler *f = nus_upscaler_create(NUS_ALG_FSR1, NUS_QUALITY_ULTRA);
    CHECK(f != NULL);
    {
        float e = -1.0f, r = -1.0f;
        CHECK(nus_upscaler_get_sharpness(f, &e, &r) == NUS_OK);
        CHECK(e == 0.0f && r > 0.79f && r < 0.81f);
    }
    nus_upscaler_destroy(f);

    nus_interp *it = nus_interp_create(NUS_WG_WIDE_32X8);
    CHECK(it != NULL);
    CHECK(nus_interp_set_input_format(it, NUS_FORMAT_BGRA8) == NUS_OK);
    {
        double ms = 0.0;
        CHECK(nus_interp_last_gpu_ms(it, &ms) != NUS_OK); /* None until an interpolation has run */
    }
    nus_interp_destroy(it);

    {
        /* table blobs are host-only */
        long long n = nus_tables_build_blob(320, 240, 640, 480, 0, NULL, 0);
        CHECK(n > 0);
    }
    {
        /* round 6: the transfer road and the diagnostics, as far as they go without a device */
        unsigned char host[64] = {0};
        nus_host_range r[4];
        long long v = -1;
        CHECK(nus_download(host, NULL, 0, NULL) == NUS_OK);                        /* nothing to move */
        CHECK(nus_download(NULL, host, sizeof host, NULL) == NUS_ERR_INVALID_ARGUMENT);
        CHECK(nus_upload(NULL, host, sizeof host, NULL) == NUS_ERR_INVALID_ARGUMENT);
        CHECK(nus_host_unpin(host) == NUS_ERR_INVALID_ARGUMENT);                   /* never pinned: the runtime is not asked */
        CHECK(strstr(nus_last_error(), "nus_host_pin") != NULL);
        CHECK(nus_host_ranges(r, 4, 0) == 0 && nus_host_ranges(NULL, 4, 1) == 0);
        CHECK(sizeof(nus_host_range) == 32);
        nus_upscaler *g = nus_upscaler_create(NUS_ALG_BICUBIC, NUS_QUALITY_QUALITY);
        CHECK(g != NULL);
        CHECK(nus_upscaler_get_option(g, "pq_narrow_active", (int64_t *)&v) == NUS_OK && v == 0); /* not initialized: no kernel yet */
        CHECK(nus_upscaler_set_option(g, "pq_narrow", 0) == NUS_OK && nus_upscaler_set_option(g, "pq_narrow", 2) == NUS_ERR_INVALID_ARGUMENT);
        CHECK(nus_upscaler_get_option(g, "no_such_option", (int64_t *)&v) == NUS_ERR_INVALID_ARGUMENT);
        CHECK(nus_upscaler_set_option(g, "inject_retire_error", 1) == NUS_ERR_INVALID_ARGUMENT); /* a test hook: NUS_TEST_HOOKS=1 only */
        nus_upscaler_destroy(g);
    }
    printf("abi_check ok\n");
    return 0;
}
