/* pss_sweep_ranks.c — a scanner sweep (pyspecsdr.py:2514-2590) sharded over ranks from plain C: one process per GPU, no Python, no torch.
 * Every rank scans its contiguous block of slices with the single-GPU entry point (pss_scan) straight into ONE packed result buffer; the
 * only exchange step is pss_gather_packed to rank 0 (RCCL over xGMI, opened by libpss.so itself).  The halo step of the display
 * accumulators (pss_halo_from_left) is shown on the per-slice peaks.
 *   gcc -O2 -D__HIP_PLATFORM_AMD__ -Iinclude -I/opt/rocm/include examples/pss_sweep_ranks.c -Lpyspecsdr_amd -lpss -L/opt/rocm/lib -lamdhip64 -lm -o /tmp/pss_sweep_ranks
 *   LD_LIBRARY_PATH=pyspecsdr_amd:/opt/rocm/lib /tmp/pss_sweep_ranks RANK N_RANKS ID_FILE [n_slices] [n] [all]
 * ("all": every rank additionally all-gathers the packed results — pss_gather_packed with dst = -1 — and prints the sum of all peaks)
 * Rendezvous: rank 0 writes the 128-byte id to ID_FILE (which must not exist beforehand), the others wait for the file.  ID_FILE "-" with N_RANKS 1: a lone rank, RCCL is
 * never opened.  Start one process per rank (any launcher: a shell loop, mpirun, srun); rank r takes device r mod pss_device_count().
 * Rank 0 prints one line per slice in sweep order: "slice k peak <dB> count <bins>"; every rank prints its halo. */
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#include "pss.h"

#define CK(call)                                                                                   \
    do {                                                                                           \
        int _r = (call);                                                                           \
        if (_r) { fprintf(stderr, "%s -> %d: %s\n", #call, _r, pss_last_error(ctx)); return 1; }   \
    } while (0)
#define HK(call)                                                                                   \
    do {                                                                                           \
        hipError_t _e = (call);                                                                    \
        if (_e != hipSuccess) { fprintf(stderr, "%s: %s\n", #call, hipGetErrorString(_e)); return 1; } \
    } while (0)

static size_t up256(size_t b) { return (b + 255) & ~(size_t)255; }

int main(int argc, char **argv)
{
    if (argc < 4) { fprintf(stderr, "usage: %s RANK N_RANKS ID_FILE [n_slices] [n]\n", argv[0]); return 2; }
    const int rank = atoi(argv[1]), n_ranks = atoi(argv[2]);
    const char *id_file = argv[3];
    const long n_slices = argc > 4 ? atol(argv[4]) : 37;
    const int n = argc > 5 ? atoi(argv[5]) : 4096;
    const int all_gather = argc > 6 && strcmp(argv[6], "all") == 0;
    const double fs = 2.4e6;
    const long halo = 3;
    pss_ctx *ctx = NULL;
    const int n_dev = pss_device_count();
    if (n_dev < 1) { fprintf(stderr, "no GPU\n"); return 2; }
    HK(hipSetDevice(rank % n_dev));
    if (pss_create(rank % n_dev, &ctx)) { fprintf(stderr, "pss_create: %s\n", pss_last_error(NULL)); return 2; }

    /* ---- rendezvous */
    if (strcmp(id_file, "-") == 0) {
        CK(pss_comm_init(ctx, NULL, rank, n_ranks));
    } else {
        unsigned char id[PSS_COMM_ID_BYTES];
        if (rank == 0) {
            CK(pss_comm_id(id));
            char tmp[4096];
            snprintf(tmp, sizeof tmp, "%s.tmp", id_file);
            FILE *f = fopen(tmp, "wb");
            if (!f || fwrite(id, 1, sizeof id, f) != sizeof id || fclose(f) || rename(tmp, id_file)) { perror("id file"); return 1; }
        } else {
            FILE *f = NULL;
            for (int tries = 0; tries < 600 && !(f = fopen(id_file, "rb")); tries++) usleep(100000);
            if (!f || fread(id, 1, sizeof id, f) != sizeof id) { fprintf(stderr, "rank %d: no id in %s\n", rank, id_file); return 1; }
            fclose(f);
        }
        CK(pss_comm_init(ctx, id, rank, n_ranks));
    }

    /* ---- this rank's block of the sweep */
    long start = 0, count = 0, cap = 0;
    CK(pss_shard_range(n_slices, rank, n_ranks, &start, &count));
    CK(pss_shard_range(n_slices, 0, n_ranks, NULL, &cap));                 /* the largest block */
    long *counts = (long *)malloc(sizeof(long) * n_ranks);
    for (int r = 0; r < n_ranks; r++) CK(pss_shard_range(n_slices, r, n_ranks, NULL, &counts[r]));
    float *iq = (float *)malloc(sizeof(float) * 2 * (size_t)n * (count > 0 ? count : 1));
    for (long k = 0; k < count; k++) {      /* slice g: a carrier at bin 16 + 29 g mod (n - 32) with amplitude rising with g, plus a weak second tone */
        const long g = start + k;
        const double f1 = (double)(16 + (29 * g) % (n - 32)) / n, f2 = (double)((7 * g) % n) / n, a1 = 0.1 + 0.01 * (double)(g % 50);
        for (int i = 0; i < n; i++) {
            iq[2 * ((size_t)k * n + i)] = (float)(a1 * cos(2.0 * M_PI * f1 * i) + 0.003 * cos(2.0 * M_PI * f2 * i));
            iq[2 * ((size_t)k * n + i) + 1] = (float)(a1 * sin(2.0 * M_PI * f1 * i) + 0.003 * sin(2.0 * M_PI * f2 * i));
        }
    }
    float *d_iq = NULL;
    HK(hipMalloc((void **)&d_iq, sizeof(float) * 2 * (size_t)n * (count > 0 ? count : 1)));
    HK(hipMemcpy(d_iq, iq, sizeof(float) * 2 * (size_t)n * count, hipMemcpyHostToDevice));

    /* ---- one packed result buffer per rank: sections [peak f32 | bandwidth f64 | count i32], each sized for the largest block */
    const size_t o_peak = 0, o_bw = up256(sizeof(float) * cap), o_cnt = o_bw + up256(sizeof(double) * cap);
    const size_t bytes = o_cnt + up256(sizeof(int) * cap);
    char *d_packed = NULL, *d_all = NULL;
    HK(hipMalloc((void **)&d_packed, bytes));
    HK(hipMemset(d_packed, 0, bytes));
    if (rank == 0) HK(hipMalloc((void **)&d_all, bytes * n_ranks));
    HK(hipDeviceSynchronize());
    CK(pss_scan(ctx, d_iq, count, n, fs, NULL, (float *)(d_packed + o_peak), (double *)(d_packed + o_bw), (int *)(d_packed + o_cnt)));
    CK(pss_gather_packed(ctx, d_packed, bytes, d_all, 0));                /* the sweep's ONE collective */

    /* ---- the halo step: the `halo` peaks before this rank's block (row = one float) */
    float *d_halo = NULL;
    long n_halo = 0;
    HK(hipMalloc((void **)&d_halo, sizeof(float) * halo));
    CK(pss_halo_from_left(ctx, d_packed + o_peak, counts, sizeof(float), halo, d_halo, &n_halo));
    CK(pss_sync(ctx));

    float h_halo[8];
    HK(hipMemcpy(h_halo, d_halo, sizeof(float) * n_halo, hipMemcpyDeviceToHost));
    printf("rank %d block %ld %ld halo", rank, start, count);
    for (long k = 0; k < n_halo; k++) printf(" %.6f", h_halo[k]);
    printf("\n");
    if (all_gather) {       /* the all-gather form of the same collective: every rank ends up with every rank's packed results */
        char *d_every = NULL;
        HK(hipMalloc((void **)&d_every, bytes * n_ranks));
        CK(pss_gather_packed(ctx, d_packed, bytes, d_every, -1));
        CK(pss_sync(ctx));
        char *every = (char *)malloc(bytes * n_ranks);
        HK(hipMemcpy(every, d_every, bytes * n_ranks, hipMemcpyDeviceToHost));
        double sum = 0.0;
        for (int r = 0; r < n_ranks; r++)
            for (long k = 0; k < counts[r]; k++) sum += ((const float *)(every + (size_t)r * bytes + o_peak))[k];
        printf("rank %d allgather %.6f\n", rank, sum);
        free(every);
        hipFree(d_every);
    }
    if (rank == 0) {
        char *all = (char *)malloc(bytes * n_ranks);
        HK(hipMemcpy(all, d_all, bytes * n_ranks, hipMemcpyDeviceToHost));
        for (int r = 0; r < n_ranks; r++) {
            long s = 0;
            CK(pss_shard_range(n_slices, r, n_ranks, &s, NULL));
            const char *blk = all + (size_t)r * bytes;
            for (long k = 0; k < counts[r]; k++)
                printf("slice %ld peak %.6f count %d bw %.3f\n", s + k, ((const float *)(blk + o_peak))[k], ((const int *)(blk + o_cnt))[k],
                       ((const double *)(blk + o_bw))[k]);
        }
        free(all);
    }
    fflush(stdout);
    hipFree(d_iq); hipFree(d_packed); hipFree(d_all); hipFree(d_halo);
    free(iq); free(counts);
    pss_destroy(ctx);
    return 0;
}
