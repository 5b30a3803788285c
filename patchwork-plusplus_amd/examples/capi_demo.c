/* capi_demo.c -- the C-ABI from plain C99 (what a cgo / JNI / ctypes-style binding sees): one KITTI-layout
 * .bin frame through pwpp_estimate_ground(), counts and the first indices printed.
 *   gcc -std=c99 -Wall -Werror -I ../../include capi_demo.c -o capi_demo -L ../lib -lpwpp_hip -Wl,-rpath,'$ORIGIN/../lib'
 *   ./capi_demo <frame.bin>
 * Mirrors the read loop of the reference demo (cpp/patchworkpp/examples/demo_visualize.cpp:18-34) and its
 * use of the object (:60-72), through the C entry points that replace each call. */
#include <stdio.h>
#include <stdlib.h>

#include "pwpp.h"

static int fail(const char *what) {
    fprintf(stderr, "%s: %s\n", what, pwpp_last_error());
    return 1;
}

int main(int argc, char **argv) {
    pwpp_params params;
    pwpp_handle *h = NULL;
    FILE *f;
    long bytes;
    int n;
    float *pts;
    int32_t n_ground = 0, n_nonground = 0, n_patches = 0;
    int32_t *ground;

    if (pwpp_params_default(&params) != PWPP_OK) return fail("pwpp_params_default"); /* Params(), patchworkpp.h:79-111 */
    params.verbose = 0;
    if (pwpp_create(&params, 0, &h) != PWPP_OK) {  /* PatchWorkpp(Params), patchworkpp.h:120 */
        /* no GPU: the library says so and does NOT fall back to a CPU path */
        fprintf(stderr, "pwpp_create: %s\n", pwpp_last_error());
        return 2;
    }
    if (argc < 2) {
        fprintf(stderr, "usage: %s <frame.bin>\n", argv[0]);
        pwpp_destroy(h);
        return 1;
    }
    f = fopen(argv[1], "rb");
    if (!f) {
        perror(argv[1]);
        return 1;
    }
    fseek(f, 0, SEEK_END);
    bytes = ftell(f);
    fseek(f, 0, SEEK_SET);
    n = (int)(bytes / (4 * (long)sizeof(float)));  /* x, y, z, intensity */
    pts = (float *)malloc((size_t)n * 4 * sizeof(float));
    if (!pts || fread(pts, 4 * sizeof(float), (size_t)n, f) != (size_t)n) {
        fprintf(stderr, "short read\n");
        return 1;
    }
    fclose(f);

    if (pwpp_estimate_ground(h, pts, n, 4, PWPP_LAYOUT_ROW_MAJOR) != PWPP_OK) return fail("pwpp_estimate_ground"); /* estimateGround(), patchworkpp.cpp:151 */
    if (pwpp_get_counts(h, 0, &n_ground, &n_nonground, &n_patches) != PWPP_OK) return fail("pwpp_get_counts");
    ground = (int32_t *)malloc((size_t)(n_ground > 0 ? n_ground : 1) * sizeof(int32_t));
    if (pwpp_get_ground_indices(h, 0, ground) != PWPP_OK) return fail("pwpp_get_ground_indices");   /* getGroundIndices(), patchworkpp.h:159 */
    printf("points %d ground %d nonground %d patches %d height %.6f time_us %.1f\n", n, (int)n_ground, (int)n_nonground,
           (int)n_patches, pwpp_get_height(h), pwpp_get_time_us(h));
    free(ground);
    pwpp_destroy(h);

    /* Batches in flight (pwpp_pipe_*, no reference counterpart): four batches of three independent frames through a pipe of depth 2.
     * From pageable host memory a submit returns with the batch done (PWPP_MEM_HOST); device or pinned buffers make it asynchronous
     * and the two handles overlap.  The handle a submit returns holds that batch until it comes round again -- and belongs to the
     * pipe: it is gone after pwpp_pipe_destroy. */
    {
        pwpp_pipe *pipe = NULL;
        const float *frames[3];
        int32_t ns[3];
        int k, same = 1, bad = 0;
        frames[0] = frames[1] = frames[2] = pts;
        ns[0] = ns[1] = ns[2] = n;
        if (pwpp_pipe_create(&params, 0, 2, &pipe) != PWPP_OK) return fail("pwpp_pipe_create");
        for (k = 0; k < 4 && !bad; ++k) {
            pwpp_handle *holder = NULL;
            int32_t g = 0, ng = 0, np = 0;
            int fr;
            if (pwpp_pipe_submit(pipe, frames, ns, 3, 4, PWPP_LAYOUT_ROW_MAJOR, PWPP_MEM_HOST, PWPP_MODE_FRESH, &holder) != PWPP_OK) bad = fail("pwpp_pipe_submit");
            if (!bad && pwpp_synchronize(holder) != PWPP_OK) bad = fail("pwpp_synchronize");
            if (!bad && holder != pwpp_pipe_handle(pipe, k % 2)) same = 0;
            for (fr = 0; fr < 3 && !bad; ++fr) {
                if (pwpp_get_counts(holder, fr, &g, &ng, &np) != PWPP_OK) bad = fail("pwpp_get_counts");
                if (g != n_ground || ng != n_nonground || np != n_patches) same = 0;
            }
        }
        if (!bad && pwpp_pipe_drain(pipe) != PWPP_OK) bad = fail("pwpp_pipe_drain");
        if (!bad) printf("pipe depth 2: 4 batches of 3 frames, every frame %s the single call\n", same ? "equal to" : "DIFFERENT FROM");

        /* The same pipe in PWPP_MODE_STREAMS (the reference's real use, demo_sequential.cpp:54-67: one long-lived object per sensor):
         * two GROUPS of two stateful streams, handle g owns group g, submit k carries the next frame of every stream of group k mod 2.
         * Every stream sees the same frame three times, so all four must end at the sensor height one handle reaches on its own. */
        if (!bad) {
            pwpp_handle *one = NULL;
            double want = 0.0;
            int step, ok = 1;
            if (pwpp_create(&params, 0, &one) != PWPP_OK) bad = fail("pwpp_create");
            for (step = 0; step < 3 && !bad; ++step)
                if (pwpp_estimate_ground(one, pts, n, 4, PWPP_LAYOUT_ROW_MAJOR) != PWPP_OK) bad = fail("pwpp_estimate_ground");
            if (!bad) want = pwpp_get_height(one);
            pwpp_destroy(one);
            if (!bad && pwpp_pipe_set_num_streams(pipe, 2) != PWPP_OK) bad = fail("pwpp_pipe_set_num_streams");
            for (k = 0; k < 6 && !bad; ++k)  /* three steps of each of the two groups */
                if (pwpp_pipe_submit(pipe, frames, ns, 2, 4, PWPP_LAYOUT_ROW_MAJOR, PWPP_MEM_HOST, PWPP_MODE_STREAMS, NULL) != PWPP_OK) bad = fail("pwpp_pipe_submit (streams)");
            if (!bad && pwpp_pipe_drain(pipe) != PWPP_OK) bad = fail("pwpp_pipe_drain");
            for (k = 0; k < 2 && !bad; ++k) {
                pwpp_state st[2];
                if (pwpp_get_state(pwpp_pipe_handle(pipe, k), 0, &st[0]) != PWPP_OK || pwpp_get_state(pwpp_pipe_handle(pipe, k), 1, &st[1]) != PWPP_OK) bad = fail("pwpp_get_state");
                else if (st[0].sensor_height != want || st[1].sensor_height != want) ok = 0;
            }
            if (!bad) printf("pipe in stream mode: 2 groups of 2 streams, 3 steps each, every stream %s one handle's sensor height %.6f\n", ok ? "at" : "AWAY FROM", want);
        }
        pwpp_pipe_destroy(pipe); /* (also on the error paths above: the pipe owns two handles and their workspaces) */
        if (bad) {
            free(pts);
            return bad;
        }
    }
    free(pts);
    return 0;
}
