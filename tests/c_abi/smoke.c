/* A plain-C99 consumer of include/jbonsai_amd.h (no ctypes in between): what a C or Rust caller's compiler sees.
 * Mirrors the reference's own first test and its streaming loop:
 *   Engine::load + Engine::synthesize, src/lib.rs:39-47      -> jb_engine_load + jb_synthesize
 *   SpeechGenerator::generate_step until it returns 0, src/speech.rs:65-96 -> jb_generator_new + jb_generator_step
 * usage: smoke VOICE.htsvoice LABELS.txt      (one full-context label per line: SAMPLE_SENTENCE_1)
 * exit: 0 ok; 77 no HIP device (the product has no CPU path); 1 anything else.
 * build: gcc -std=c99 -Wall -Wextra -Werror -pedantic -Iinclude tests/c_abi/smoke.c -Ljbonsai_amd -ljbonsai_amd -lm */
#include "jbonsai_amd.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define MAX_LINES 256

static int fail(const char *what, int rc)
{
    fprintf(stderr, "smoke.c: %s failed (%d): %s\n", what, rc, jb_last_error());
    return 1;
}

int main(int argc, char **argv)
{
    static char text[1 << 16];
    const char *lines[MAX_LINES];
    size_t n_lines = 0, n_read, n_samples = 0, got = 0, fperiod;
    const char *voice;
    jb_engine *eng = NULL;
    jb_generator *gen = NULL;
    double *pcm = NULL, *stream = NULL, max_diff = 0.0;
    char arch[32];
    FILE *f;
    char *p;
    long r;
    int rc;
    size_t i;

    if (argc != 3) {
        fprintf(stderr, "usage: %s VOICE.htsvoice LABELS.txt\n", argv[0]);
        return 1;
    }
    printf("%s, %d device(s)\n", jb_version(), jb_device_count());
    if (jb_device_count() <= 0) {
        fprintf(stderr, "smoke.c: no HIP device (the product has no CPU path)\n");
        return 77;
    }
    if (jb_device_arch(0, arch, sizeof arch) == JB_OK)
        printf("device 0: %s\n", arch);
    f = fopen(argv[2], "rb");
    if (!f)
        return fail("fopen(labels)", 0);
    n_read = fread(text, 1, sizeof text - 1, f);
    fclose(f);
    text[n_read] = '\0';
    for (p = strtok(text, "\n"); p && n_lines < MAX_LINES; p = strtok(NULL, "\n"))
        if (*p)
            lines[n_lines++] = p;

    voice = argv[1];
    if ((rc = jb_engine_load(&voice, 1, &eng)) != JB_OK)
        return fail("jb_engine_load", rc);
    if (jb_engine_get_sampling_frequency(eng) != 48000 || jb_engine_get_fperiod(eng) != 240 ||
        jb_engine_num_voices(eng) != 1 || jb_engine_num_streams(eng) != 3 || jb_engine_num_states(eng) != 5)
        return fail("the nitech voice's metadata (src/model/mod.rs:214-232)", 0);

    /* Engine::synthesize: length and the two samples src/lib.rs:44-46 pins */
    if ((rc = jb_synthesize(eng, lines, n_lines, &pcm, &n_samples)) != JB_OK)
        return fail("jb_synthesize", rc);
    printf("jb_synthesize: %lu labels -> %lu samples, [2000] = %.12f, [30000] = %.10f\n", (unsigned long)n_lines,
           (unsigned long)n_samples, n_samples > 2000 ? pcm[2000] : 0.0, n_samples > 30000 ? pcm[30000] : 0.0);
    if (n_samples != 66480)
        return fail("length 66480 (src/lib.rs:44)", (int)n_samples);
    if (fabs(pcm[2000] - 19.35141137623778) > 1e-6 || fabs(pcm[30000] - -980.6757547598129) > 1e-6)
        return fail("golden samples (src/lib.rs:45-46)", 0);

    /* the streaming iterator over the same labels: fperiod samples per step, 0 at the end */
    if ((rc = jb_generator_new(eng, lines, n_lines, &gen)) != JB_OK)
        return fail("jb_generator_new", rc);
    fperiod = jb_generator_fperiod(gen);
    if (fperiod != 240 || jb_generator_total_frames(gen) != 277)
        return fail("generator geometry", (int)fperiod);
    stream = (double *)malloc(sizeof(double) * n_samples);
    if (!stream)
        return fail("malloc", 0);
    if (jb_generator_step(gen, stream, fperiod - 1) != JB_ERR_BUFFER) /* src/speech.rs:69-71: a panic there */
        return fail("short buffer must be JB_ERR_BUFFER", 0);
    while ((r = jb_generator_step(gen, stream + got, n_samples - got)) > 0)
        got += (size_t)r;
    if (r < 0)
        return fail("jb_generator_step", (int)r);
    if (got != n_samples || jb_generator_synthesized_frames(gen) != 277)
        return fail("generator length", (int)got);
    for (i = 0; i < n_samples; i++) {
        const double d = fabs(stream[i] - pcm[i]);
        if (d > max_diff)
            max_diff = d;
    }
    printf("jb_generator_step: %lu samples, max |stream - synthesize| = %.3e\n", (unsigned long)got, max_diff);
    if (max_diff > 1e-5 || fabs(stream[30000] - -980.6757547598129) > 1e-6)
        return fail("generator samples", 0);

    free(stream);
    jb_generator_free(gen);
    jb_pcm_free(pcm);
    jb_engine_free(eng);
    printf("ok\n");
    return 0;
}
