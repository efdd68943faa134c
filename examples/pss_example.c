/* pss_example.c — libpss.so from plain C (no Python, no torch): one read buffer through compute_fft, measure_signal_power and
 * demodulate_signal(NFM), the three calls of the reference's main loop (pyspecsdr.py:2251, 2262, 2275).
 *   gcc -O2 -Iinclude examples/pss_example.c -Lpyspecsdr_amd -lpss -lm -o /tmp/pss_example
 *   LD_LIBRARY_PATH=pyspecsdr_amd:/opt/rocm/lib /tmp/pss_example [n] [fs]
 * Prints the values a caller would look at first; tests/test_gpu_parity.py compares them with the Python shim's. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include "pss.h"

int main(int argc, char **argv)
{
    const int n = argc > 1 ? atoi(argv[1]) : 1024;
    const double fs = argc > 2 ? atof(argv[2]) : 2.4e6;
    pss_ctx *ctx = NULL;
    if (pss_create(0, &ctx)) { fprintf(stderr, "pss_create: %s\n", pss_last_error(NULL)); return 2; }
    float *iq = (float *)malloc(sizeof(float) * 2 * n);
    double ph = 0.0;
    for (int i = 0; i < n; i++) { /* FM: 1 kHz tone, 5 kHz deviation (deterministic, no RNG) */
        ph += 2.0 * M_PI * 5e3 * sin(2.0 * M_PI * 1000.0 * i / fs) / fs;
        iq[2 * i] = (float)(0.5 * cos(ph));
        iq[2 * i + 1] = (float)(0.5 * sin(ph));
    }
    double *db = (double *)malloc(sizeof(double) * n);
    float power = 0.0f;
    int rc = pss_h_compute_fft(ctx, iq, n, db);
    if (!rc) rc = pss_h_measure_power(ctx, iq, n, &power);
    const int n_out = pss_demod_out_len(PSS_MODE_NFM, n, fs);
    double *audio = (double *)malloc(sizeof(double) * 2 * (n_out > 0 ? n_out : 1));
    int16_t *pcm = (int16_t *)malloc(sizeof(int16_t) * 2 * (n_out > 0 ? n_out : 1));
    if (!rc) rc = pss_h_demodulate(ctx, PSS_MODE_NFM, iq, n, fs, audio, pcm);
    if (rc) { fprintf(stderr, "libpss error %d: %s\n", rc, pss_last_error(ctx)); return 1; }
    int kmax = 0;
    for (int k = 1; k < n; k++) if (db[k] > db[kmax]) kmax = k;
    printf("n %d n_out %d power_db %.6f peak_bin %d peak_db %.6f\n", n, n_out, power, kmax, db[kmax]);
    printf("pcm");
    for (int k = 0; k < n_out && k < 10; k++) printf(" %d", pcm[2 * k]);
    printf("\n");
    free(iq); free(db); free(audio); free(pcm);
    pss_destroy(ctx);
    return 0;
}
