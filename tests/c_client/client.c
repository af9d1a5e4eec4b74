/* A C program on the C ABI alone (include/sarx.h + libsarx.so, no Python, no torch): focuses a seeded
 * point-target scene with sarx_csa_focus_host, forms the ATI/DPCA products of the image with itself, and
 * prints numbers the pytest wrapper (tests/test_gpu_c_client.py) checks against the oracle.
 *   gcc -O2 -I include tests/c_client/client.c -o tests/c_client/client -L <dir of libsarx.so> -lsarx -lm
 * usage: client <in.bin> <n_az> <n_rg> <8 radar doubles...> <out.bin>
 * in.bin / out.bin: row-major complex64 [n_az][n_rg]. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "sarx.h"

static void die(sarx_ctx* ctx, const char* what) {
    fprintf(stderr, "%s: %s\n", what, sarx_last_error(ctx));
    exit(2);
}

int main(int argc, char** argv) {
    if (argc != 13) { fprintf(stderr, "usage: client in n_az n_rg p0..p7 out\n"); return 1; }
    const int n_az = atoi(argv[2]), n_rg = atoi(argv[3]);
    sarx_radar_params prm;
    double* pv = (double*)&prm;
    for (int i = 0; i < 8; ++i) pv[i] = atof(argv[4 + i]);
    const size_t n = (size_t)n_az * n_rg;
    float* in = malloc(n * 8);
    float* out = malloc(n * 8);
    FILE* f = fopen(argv[1], "rb");
    if (!f || fread(in, 8, n, f) != n) { fprintf(stderr, "cannot read %s\n", argv[1]); return 1; }
    fclose(f);

    sarx_ctx* ctx = NULL;
    sarx_plan* plan = NULL;
    if (sarx_init(0, &ctx)) die(NULL, "sarx_init");
    if (sarx_csa_plan_create(ctx, n_az, n_rg, &prm, SARX_FUSE_RANGE, &plan)) die(ctx, "plan_create");
    if (sarx_csa_focus_host(plan, in, out)) die(ctx, "focus_host");

    /* errors come back as codes with text, never as a fallback */
    sarx_plan* bad = NULL;
    const int rc = sarx_csa_plan_create(ctx, 1, n_rg, &prm, 0, &bad);
    printf("bad_plan_rc %d msg \"%s\"\n", rc, sarx_last_error(ctx));

    /* round 4: the same image again through the entry points a frame loop uses - a page-locked result buffer (sarx_host_alloc: the
     * download is one DMA) and two frames in flight on two lanes, each lane with its own plan and device buffers */
    {
        void *pin = NULL, *d_in = NULL, *d_img[2] = {NULL, NULL};
        sarx_plan* plan1 = NULL;
        if (sarx_host_alloc(ctx, n * 8, &pin)) die(ctx, "host_alloc");
        if (sarx_csa_plan_create(ctx, n_az, n_rg, &prm, SARX_FUSE_RANGE, &plan1)) die(ctx, "plan_create (lane 1)");
        if (sarx_malloc(ctx, n * 8, &d_in) || sarx_malloc(ctx, n * 8, &d_img[0]) || sarx_malloc(ctx, n * 8, &d_img[1])) die(ctx, "malloc");
        if (sarx_memcpy_h2d(ctx, d_in, in, n * 8)) die(ctx, "h2d");
        if (sarx_set_range_cus(ctx, 192)) die(ctx, "set_range_cus");
        for (int f = 0; f < 4; ++f) {
            if (sarx_select_lane(ctx, f & 1)) die(ctx, "select_lane");
            if (sarx_csa_focus_dev((f & 1) ? plan1 : plan, d_in, d_img[f & 1])) die(ctx, "focus_dev");
        }
        if (sarx_select_lane(ctx, 0) || sarx_set_range_cus(ctx, 0) || sarx_lanes_join(ctx) || sarx_sync(ctx)) die(ctx, "join");
        int same = 1;
        for (int k = 0; k < 2; ++k) {
            if (sarx_memcpy_d2h(ctx, pin, d_img[k], n * 8)) die(ctx, "d2h");
            same = same && memcmp(pin, out, n * 8) == 0;
        }
        printf("lanes_bit_identical %d lane_out_of_range_rc %d\n", same, sarx_select_lane(ctx, 9));
        sarx_free(ctx, d_in); sarx_free(ctx, d_img[0]); sarx_free(ctx, d_img[1]);
        sarx_csa_plan_destroy(plan1);
        sarx_host_free(ctx, pin);
    }

    /* round 5: the host-array call as a pipeline - frame i+1 uploads while frame i focuses and downloads (sarx_csa_focus_host_begin /
     * _end), once into page-locked results (asynchronous DMA) and once into ordinary memory (downloaded by _end); a third pending frame
     * on one plan is refused with an error code */
    {
        void* pin[2] = {NULL, NULL};
        float* pageable = malloc(n * 8);
        int t0 = -1, t1 = -1, t2 = -1, same = 1;
        if (sarx_host_alloc(ctx, n * 8, &pin[0]) || sarx_host_alloc(ctx, n * 8, &pin[1])) die(ctx, "host_alloc");
        if (sarx_csa_focus_host_begin(plan, in, pin[0], &t0)) die(ctx, "host_begin 0");
        if (sarx_csa_focus_host_begin(plan, in, pin[1], &t1)) die(ctx, "host_begin 1");
        const int rc3 = sarx_csa_focus_host_begin(plan, in, pageable, &t2);
        if (sarx_csa_focus_host_end(plan, t0)) die(ctx, "host_end 0");
        if (sarx_csa_focus_host_begin(plan, in, pageable, &t2)) die(ctx, "host_begin 2");
        if (sarx_csa_focus_host_end(plan, t1) || sarx_csa_focus_host_end(plan, t2)) die(ctx, "host_end");
        same = memcmp(pin[0], out, n * 8) == 0 && memcmp(pin[1], out, n * 8) == 0 && memcmp(pageable, out, n * 8) == 0;
        printf("pipeline_bit_identical %d third_pending_rc %d end_twice_rc %d\n", same, rc3, sarx_csa_focus_host_end(plan, t0));
        sarx_host_free(ctx, pin[0]); sarx_host_free(ctx, pin[1]);
        free(pageable);
    }

    /* range axis and cross-range axis as the reference returns them */
    double* rax = malloc(sizeof(double) * n_rg);
    double* cax = malloc(sizeof(double) * n_az);
    if (sarx_csa_axes(plan, rax, cax)) die(ctx, "axes");
    printf("range_axis %.9f %.9f cross_range %.9f %.9f\n", rax[0], rax[n_rg - 1], cax[0], cax[n_az - 1]);

    f = fopen(argv[12], "wb");
    if (!f || fwrite(out, 8, n, f) != n) { fprintf(stderr, "cannot write %s\n", argv[12]); return 1; }
    fclose(f);
    sarx_csa_plan_destroy(plan);
    sarx_destroy(ctx);
    free(in); free(out); free(rax); free(cax);
    printf("ok\n");
    return 0;
}
