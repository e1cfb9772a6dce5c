/* ask_greedy.c — the greedy generation loop of cmd/ask (generateResponse, cmd/ask/main.go:287-360) written against the C
 * ABI alone (include/nvllm.h): what the cgo shim of INTEGRATION.md does, without Go and without Python.
 *
 *   ask_greedy <model-dir> <max-new-tokens> <token id> [<token id> ...]
 *
 * <model-dir> holds config.json (or model_info.json) and model.safetensors (or the sharded index): the layout
 * scripts/download_model.py writes for the reference.  Tokenisation is out of scope (the reference shells out to a
 * Python tokenizer, cmd/ask/main.go:362-387): prompt and output are token ids.
 * build: gcc -O2 -I include examples/ask_greedy.c -L nano-vllm-go_amd/lib -lnvllm_hip -Wl,-rpath,... -o ask_greedy */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "nvllm.h"

static int die(nvl_model* m, const char* what) {
    fprintf(stderr, "%s: %s\n", what, nvl_last_error(m));
    if (m) nvl_destroy(m);
    return 1;
}

int main(int argc, char** argv) {
    if (argc < 4) {
        fprintf(stderr, "usage: %s <model-dir> <max-new-tokens> <token id> [...]\n", argv[0]);
        return 2;
    }
    const char* dir = argv[1];
    const int max_new = atoi(argv[2]);
    const int n_prompt = argc - 3;
    if (max_new <= 0) { fprintf(stderr, "max-new-tokens must be positive\n"); return 2; }
    int32_t* prompt = (int32_t*)malloc(sizeof(int32_t) * (size_t)n_prompt);
    for (int i = 0; i < n_prompt; i++) prompt[i] = (int32_t)atoi(argv[3 + i]);

    char path[4096];
    nvl_model_config cfg;
    snprintf(path, sizeof path, "%s/config.json", dir);
    if (nvl_load_config_json(path, &cfg) != NVL_OK) {                 /* LoadModelFromDirectory, generic_loader.go:1016-1027 */
        snprintf(path, sizeof path, "%s/model_info.json", dir);
        if (nvl_load_config_json(path, &cfg) != NVL_OK) return die(NULL, "config");
    }
    nvl_runtime_opts opts;
    memset(&opts, 0, sizeof opts);
    opts.device = 0;
    opts.precision = getenv("NVL_F32") ? NVL_PRECISION_F32 : NVL_PRECISION_BF16;
    opts.max_seqs = 1;
    opts.max_batch_tokens = n_prompt > 64 ? n_prompt : 64;
    opts.tp_size = 1;
    nvl_model* m = NULL;
    if (nvl_create(&cfg, &opts, &m) != NVL_OK) return die(NULL, "nvl_create");
    if (nvl_load_safetensors(m, dir) != NVL_OK) return die(m, "nvl_load_safetensors");
    if (nvl_finalize(m) != NVL_OK) return die(m, "nvl_finalize");

    /* prefill (main.go:293: ForwardWithCache(prompt, kv, 0)) -> first token by argmax (main.go:389-402) */
    const int64_t seq = 0;
    const int32_t len = n_prompt, pos0 = 0;
    int32_t first = 0;
    if (nvl_seq_open(m, seq) != NVL_OK) return die(m, "nvl_seq_open");
    if (nvl_forward(m, 1, &seq, prompt, &len, &pos0, 0, NULL, &first) != NVL_OK) return die(m, "nvl_forward");
    printf("%d", first);
    /* decode loop (main.go:315-360) as one call: the token feedback stays on the device */
    if (max_new > 1) {
        int32_t* out = (int32_t*)malloc(sizeof(int32_t) * (size_t)(max_new - 1));
        if (nvl_decode_greedy(m, 1, &seq, &first, max_new - 1, out) != NVL_OK) return die(m, "nvl_decode_greedy");
        for (int i = 0; i < max_new - 1; i++) printf(" %d", out[i]);
        free(out);
    }
    printf("\n");
    nvl_stats st;
    if (nvl_get_stats(m, &st) == NVL_OK && st.prefill_ms > 0 && st.decode_ms > 0)      /* the prints of main.go:196-198 */
        fprintf(stderr, "prefill %.1f tok/s, decode %.1f tok/s\n", 1e3 * (double)st.prefill_tokens / st.prefill_ms,
                1e3 * (double)st.decode_tokens / st.decode_ms);
    nvl_destroy(m);
    free(prompt);
    return 0;
}
