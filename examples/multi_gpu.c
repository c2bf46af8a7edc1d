/*
 * multi_gpu.c — the multi-GPU seam of the C ABI from plain C: one process per GPU, the frame sharded by 64x64 pixel
 * tile, the Film tiles gathered to rank 0 over RCCL / xGMI.  No Python, no torch, no MPI: the 128-byte communicator id
 * travels through a pipe.
 *
 * This is what replaces the reference's worker threads adding tiles into one Mutex<Vec<f32>>
 * (src/bin/craytracer.rs:245, 271-291): rank 0 alone runs the host side (Scene::new, upload) and replicates the scene
 * into the other GPUs' HBM with cray_scene_broadcast; every rank renders all samples of its tiles; one gather.
 *
 *   gcc -std=c11 -Iinclude examples/multi_gpu.c -Lcraytracer_amd/csrc -lcray_hip -Wl,-rpath,$PWD/craytracer_amd/csrc -lm -o multi_gpu
 *   ./multi_gpu 8 out.exr        # 8 ranks on GPUs 0..7 of this node
 */
#define _POSIX_C_SOURCE 200809L
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/types.h>
#include <sys/wait.h>
#include <unistd.h>

#include "cray.h"
#include "cray_host.h"
#include "cray_io.h"
#include "example_scene.h"

#define CHECK(call)                                                              \
    do {                                                                         \
        int rc_ = (call);                                                        \
        if (rc_ != CRAY_OK) { fprintf(stderr, "rank %d: %s failed (%d): %s\n", rank, #call, rc_, cray_last_error()); return 1; } \
    } while (0)

static int run_rank(int rank, int world, const cray_comm_id* id, const char* out_path) {
    cray_ctx* ctx = NULL;
    /* GPU = local rank; CRAY_ONE_DEVICE=1 puts every rank on GPU 0 (rehearsals on a one-GPU box with a stand-in collective
     * library, CRAY_RCCL_LIB: real RCCL refuses two ranks on one device) */
    CHECK(cray_ctx_create(getenv("CRAY_ONE_DEVICE") ? 0 : rank, NULL, &ctx));
    CHECK(cray_comm_init(ctx, id, rank, world));

    /* the host side of the reference (parse, Scene::new) runs once, on rank 0 */
    example_scene ex;
    cray_host_scene* host = NULL;
    cray_scene* mine = NULL;
    if (rank == 0) {
        example_scene_init(&ex);
        CHECK(cray_host_scene_new_on(&ex.desc, CRAY_SPLIT_SAH, ctx, &host));
        CHECK(cray_scene_upload(ctx, cray_host_scene_flat(host), &mine));
    }
    cray_scene* scene = NULL;
    CHECK(cray_scene_broadcast(ctx, mine, 0, &scene));   /* HBM of rank 0 -> HBM of every rank */

    uint32_t w = 0, h = 0;
    cray_scene_info(scene, &w, &h, NULL, NULL);
    cray_render_params prm;
    cray_render_params_default(&prm);   /* rank / world_size come from the communicator */
    prm.seed = 2;
    const size_t n = (size_t)w * h * 3;
    float* film = rank == 0 ? (float*)malloc(n * sizeof(float)) : NULL;
    cray_stats st;
    CHECK(cray_comm_barrier(ctx));
    CHECK(cray_render_gather(ctx, scene, &prm, film, &st));

    /* whole-job figures: sum of the rays, max of the times */
    double sums[2] = {(double)st.paths, (double)(st.closest_rays + st.shadow_rays)}, t = st.seconds;
    CHECK(cray_comm_allreduce_f64(ctx, sums, 2, CRAY_REDUCE_SUM));
    CHECK(cray_comm_allreduce_f64(ctx, &t, 1, CRAY_REDUCE_MAX));
    if (rank == 0) {
        CHECK(cray_write_exr(out_path, w, h, film));
        double mean = 0.0;
        for (size_t i = 0; i < n; i++) mean += film[i];
        printf("%d rank(s): %.0f paths, %.0f rays, mean radiance %.9g, %.3f ms -> %s\n", world, sums[0], sums[1], mean / (double)n, t * 1e3, out_path);
    }
    free(film);
    cray_scene_free(scene);
    if (host) cray_host_scene_free(host);
    cray_ctx_destroy(ctx);
    return 0;
}

int main(int argc, char** argv) {
    const int world = argc > 1 ? atoi(argv[1]) : 1;
    const char* out_path = argc > 2 ? argv[2] : "multi_gpu.exr";
    if (world < 1 || world > 64) { fprintf(stderr, "usage: %s <ranks> [out.exr]\n", argv[0]); return 2; }

    /* fork BEFORE anything touches the GPU; the children wait for the id on their pipe */
    int (*pipes)[2] = malloc(sizeof(int[2]) * (size_t)world);
    pid_t* kids = malloc(sizeof(pid_t) * (size_t)world);
    for (int r = 1; r < world; r++) {
        if (pipe(pipes[r]) != 0) { perror("pipe"); return 1; }
        kids[r] = fork();
        if (kids[r] < 0) { perror("fork"); return 1; }
        if (kids[r] == 0) {
            close(pipes[r][1]);
            cray_comm_id id;
            size_t got = 0;
            while (got < sizeof(id)) {
                ssize_t k = read(pipes[r][0], (char*)&id + got, sizeof(id) - got);
                if (k <= 0) { fprintf(stderr, "rank %d: no communicator id\n", r); _exit(1); }
                got += (size_t)k;
            }
            _exit(run_rank(r, world, &id, out_path));
        }
        close(pipes[r][0]);
    }
    int rank = 0;
    cray_comm_id id;
    CHECK(cray_comm_unique_id(&id));
    for (int r = 1; r < world; r++)
        if (write(pipes[r][1], &id, sizeof(id)) != (ssize_t)sizeof(id)) { perror("write"); return 1; }
    int rc = run_rank(0, world, &id, out_path);
    for (int r = 1; r < world; r++) {
        int status = 0;
        waitpid(kids[r], &status, 0);
        if (!WIFEXITED(status) || WEXITSTATUS(status) != 0) rc = 1;
    }
    free(pipes);
    free(kids);
    return rc;
}
