// TEST TOOL: concurrent generate() calls through the C ABI on the emulator library, built for ThreadSanitizer (tools/emu_tsan.sh).
// No Python in the process: TSan's interceptors and the interpreter do not get along, and the code under
// test — csrc/engine.hip: the decode pool's driver thread, request queue, row bookkeeping, pool (re)creation, session forks — is
// reached entirely through include/vcoder_hip.h.  Input: the file tools/emu_tsan/prepare.py writes.
//   - N session threads, each K vc_generate / vc_generate_greedy calls with varying prompts, lengths, EOS and sampling
//   - every result is compared with the same call made alone afterwards (the pool must not change a request's ids)
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>

#include <string>
#include <thread>
#include <vector>

#include "vcoder_hip.h"

struct Prompt {
    int B, T, seg, dep;
    std::vector<int64_t> ids;
};
struct Call {
    int prompt, max_new, eos, sample;
    uint64_t seed;
    std::vector<int32_t> out;
    int n = 0, rc = 0;
};

#define CK(x)                                                                                   \
    do {                                                                                        \
        int rc_ = (x);                                                                          \
        if (rc_ < 0) {                                                                          \
            fprintf(stderr, "%s failed: %d (%s)\n", #x, rc_, vc_last_error(ctx0));             \
            return 1;                                                                           \
        }                                                                                       \
    } while (0)

static int run_call(vc_model* m, const Prompt& p, const std::vector<float>& px, Call& c) {
    const size_t per = (size_t)3 * (px.size() / 3 / 3);   // unused; pixels are one block of 3 images per modality below
    (void)per;
    const float* img = px.data();
    const float* seg = p.seg ? px.data() + px.size() / 3 : nullptr;
    const float* dep = p.dep ? px.data() + 2 * (px.size() / 3) : nullptr;
    c.out.assign((size_t)p.B * c.max_new, -7);
    if (c.sample) {
        vc_sampling s{1, 0.9f, 20, 0.95f, c.seed};
        return vc_generate(m, p.ids.data(), p.B, p.T, img, seg, dep, 0, c.max_new, c.eos, 0, nullptr, nullptr, 0, &s, nullptr, nullptr, 1,
                           c.out.data(), &c.n);
    }
    return vc_generate_greedy(m, p.ids.data(), p.B, p.T, img, seg, dep, 0, c.max_new, c.eos, 0, c.out.data(), &c.n);
}

int main(int argc, char** argv) {
    if (argc < 2) return 2;
    FILE* f = fopen(argv[1], "r");
    if (!f) return 2;
    vc_model_cfg cfg;
    int iv[21];
    float fv[3];
    {   // 21 fields in struct order; three of them floats
        double d[21];
        for (int i = 0; i < 21; ++i)
            if (fscanf(f, "%lf", &d[i]) != 1) return 2;
        int k = 0;
        cfg.variant = (int)d[k++]; cfg.vit_hidden = (int)d[k++]; cfg.vit_heads = (int)d[k++]; cfg.vit_ffn = (int)d[k++];
        cfg.vit_layers = (int)d[k++]; cfg.vit_layers_used = (int)d[k++]; cfg.vit_image = (int)d[k++]; cfg.vit_patch = (int)d[k++];
        cfg.vit_keep_cls = (int)d[k++]; cfg.vit_ln_eps = (float)d[k++]; cfg.hidden = (int)d[k++]; cfg.heads = (int)d[k++];
        cfg.ffn = (int)d[k++]; cfg.layers = (int)d[k++]; cfg.vocab = (int)d[k++]; cfg.max_positions = (int)d[k++];
        cfg.rms_eps = (float)d[k++]; cfg.rope_theta = (float)d[k++]; cfg.mm_proj_depth = (int)d[k++]; cfg.seg_proj_depth = (int)d[k++];
        cfg.pad_token_id = (int)d[k++];
        (void)iv; (void)fv;
    }
    vc_ctx* ctx0 = nullptr;
    vc_model* root = nullptr;
    if (vc_init(0, &ctx0) != 0) return 1;
    CK(vc_model_create(ctx0, &cfg, &root));
    int nspec = 0;
    if (fscanf(f, "%d", &nspec) != 1) return 2;
    for (int i = 0; i < nspec; ++i) {
        char key[512];
        int nd = 0;
        int64_t shape[8];
        unsigned seed;
        double off, hw;
        if (fscanf(f, "%511s %d", key, &nd) != 2) return 2;
        for (int d = 0; d < nd; ++d) {
            long long v;
            if (fscanf(f, "%lld", &v) != 1) return 2;
            shape[d] = v;
        }
        if (fscanf(f, "%u %lf %lf", &seed, &off, &hw) != 3) return 2;
        CK(vc_model_synth_tensor(root, key, shape, nd, seed, (float)off, (float)hw));
    }
    CK(vc_model_finalize(root));
    int np = 0;
    if (fscanf(f, "%d", &np) != 1) return 2;
    std::vector<Prompt> prompts(np);
    for (auto& p : prompts) {
        if (fscanf(f, "%d %d %d %d", &p.B, &p.T, &p.seg, &p.dep) != 4) return 2;
        p.ids.resize((size_t)p.B * p.T);
        for (auto& t : p.ids) {
            long long v;
            if (fscanf(f, "%lld", &v) != 1) return 2;
            t = v;
        }
    }
    int S = 0;
    if (fscanf(f, "%d", &S) != 1) return 2;
    fclose(f);
    // pixels: 2 samples x 3 modalities, any smooth non-zero pattern (depth must not be all zero: the reference's sentinel)
    std::vector<float> px((size_t)3 * 2 * 3 * S * S);
    for (size_t i = 0; i < px.size(); ++i) px[i] = 0.5f * sinf(0.013f * (float)i) + 0.1f;

    const int NS = argc > 2 ? atoi(argv[2]) : 4, K = argc > 3 ? atoi(argv[3]) : 3;
    std::vector<vc_ctx*> ctxs(NS, nullptr);
    std::vector<vc_model*> sess(NS, nullptr);
    sess[0] = root;
    ctxs[0] = ctx0;
    for (int s = 1; s < NS; ++s) {
        if (vc_init(0, &ctxs[s]) != 0) return 1;
        CK(vc_model_create_shared(ctxs[s], root, &sess[s]));
    }
    std::vector<std::vector<Call>> plan(NS, std::vector<Call>(K));
    unsigned lcg = 12345;
    auto rnd = [&]() { lcg = lcg * 1664525u + 1013904223u; return lcg >> 8; };
    for (int s = 0; s < NS; ++s)
        for (int k = 0; k < K; ++k) {
            Call& c = plan[s][k];
            c.prompt = (int)(rnd() % (unsigned)np);
            c.max_new = 2 + (int)(rnd() % 8);
            c.eos = (rnd() % 4 == 0) ? (int)(3 + rnd() % 200) : -1;
            c.sample = rnd() % 3 == 0;
            c.seed = rnd();
        }
    std::vector<std::thread> th;
    for (int s = 0; s < NS; ++s)
        th.emplace_back([&, s]() {
            for (int k = 0; k < K; ++k) plan[s][k].rc = run_call(sess[s], prompts[plan[s][k].prompt], px, plan[s][k]);
        });
    for (auto& t : th) t.join();
    int bad = 0;
    for (int s = 0; s < NS; ++s)
        for (int k = 0; k < K; ++k) {
            Call& c = plan[s][k];
            if (c.rc < 0) {
                fprintf(stderr, "session %d call %d failed: %d (%s)\n", s, k, c.rc, vc_last_error(ctxs[s]));
                ++bad;
                continue;
            }
            Call lone = c;
            lone.rc = run_call(root, prompts[c.prompt], px, lone);
            if (lone.rc < 0 || lone.n != c.n || lone.out != c.out) {
                fprintf(stderr, "session %d call %d: pooled ids differ from the lone call (n %d vs %d)\n", s, k, c.n, lone.n);
                ++bad;
            }
        }
    for (int s = 1; s < NS; ++s) {
        vc_model_destroy(sess[s]);
        vc_shutdown(ctxs[s]);
    }
    vc_model_destroy(root);
    vc_shutdown(ctx0);
    printf("stress: %d sessions x %d calls, %d mismatching\n", NS, K, bad);
    return bad ? 1 : 0;
}
