// C-ABI group A: tiny-cuda-nn module boundary (include/nerfvo_hip.h).  Host-side objects only;
// all device work is in grid.hip / mlp.hip / sh.hip.
#include "nvo_kernels.h"
#include "../../include/nerfvo_hip.h"

#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <map>
#include <memory>
#include <string>
#include <vector>

// ---------------------------------------------------------------------------------------------
// error string
// ---------------------------------------------------------------------------------------------
static thread_local char g_err[1024] = "";

void nvo_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
extern "C" const char* nvo_last_error(void) { return g_err; }
extern "C" int nvo_version(void) { return 100; }

// ---------------------------------------------------------------------------------------------
// graph-capture-safe growable scratch (nvo_common.h)
// ---------------------------------------------------------------------------------------------
int nvo_scratch_reserve(NvoScratch* s, size_t need, hipStream_t stream, const char* what) {
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    // (the legacy NULL stream cannot be captured, and querying it while ANOTHER stream captures is an error)
    if (stream != nullptr) NVO_CHECK_HIP(hipStreamIsCapturing(stream, &cs));
    const bool capturing = cs != hipStreamCaptureStatusNone;
    if (need > s->bytes) {
        NVO_REQUIRE(!capturing,
                    "%s: scratch would have to grow from %zu to %zu bytes while a hipGraph is being captured "
                    "(hipMalloc cannot be captured): launch once eagerly at this batch size before capturing",
                    what, s->bytes, need);
        if (s->ptr) {
            if (s->captured) {  // a captured graph still addresses the old block: keep it alive
                NVO_REQUIRE(s->n_retired < 8, "%s: scratch grew %u times after a graph capture", what, s->n_retired);
                s->retired[s->n_retired++] = s->ptr;
            } else {
                NVO_CHECK_HIP(hipFree(s->ptr));
            }
        }
        s->ptr = nullptr;
        s->bytes = 0;
        s->captured = false;
        NVO_CHECK_HIP(hipMalloc(&s->ptr, need));
        s->bytes = need;
    }
    if (capturing) s->captured = true;
    return NVO_OK;
}

void nvo_scratch_destroy(NvoScratch* s) {
    if (s->ptr) (void)hipFree(s->ptr);
    for (uint32_t i = 0; i < s->n_retired; ++i) (void)hipFree(s->retired[i]);
    *s = NvoScratch();
}

// ---------------------------------------------------------------------------------------------
// per-launch HIP-event profiler: events are recorded on the stream each launcher enqueues on, so
// the elapsed time is that kernel's (plus its memsets') device time, not host time.
// ---------------------------------------------------------------------------------------------
namespace {
struct ProfRec {
    char name[56];
    hipEvent_t a, b;
    bool closed;
};
bool g_prof_on = false;
std::vector<ProfRec> g_prof;
std::vector<hipEvent_t> g_prof_pool;

hipEvent_t prof_event() {
    if (!g_prof_pool.empty()) {
        hipEvent_t e = g_prof_pool.back();
        g_prof_pool.pop_back();
        return e;
    }
    hipEvent_t e = nullptr;
    (void)hipEventCreate(&e);
    return e;
}
}  // namespace

bool nvo_prof_detail() {
    static const bool on = [] { const char* e = getenv("NVO_PROF_DETAIL"); return e && atoi(e) != 0; }();
    return on;
}
static int g_prof_mute = 0;
void nvo_prof_mute(int delta) { g_prof_mute += delta; }
bool nvo_prof_enabled() { return g_prof_on && (g_prof_mute == 0 || nvo_prof_detail()); }

void nvo_prof_begin(hipStream_t s, const char* fmt, ...) {
    ProfRec r;
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(r.name, sizeof(r.name), fmt, ap);
    va_end(ap);
    r.a = prof_event();
    r.b = prof_event();
    r.closed = false;
    (void)hipEventRecord(r.a, s);
    g_prof.push_back(r);
}

void nvo_prof_end(hipStream_t s) {
    for (size_t i = g_prof.size(); i-- > 0;) {
        if (!g_prof[i].closed) {
            (void)hipEventRecord(g_prof[i].b, s);
            g_prof[i].closed = true;
            return;
        }
    }
}

extern "C" int nvo_profile_enable(int on) {
    for (auto& r : g_prof) {
        g_prof_pool.push_back(r.a);
        g_prof_pool.push_back(r.b);
    }
    g_prof.clear();
    g_prof_on = on != 0;
    return NVO_OK;
}

// Writes "name,launches,total_ms\n" lines (aggregated by name) into buf; returns bytes needed.
extern "C" int64_t nvo_profile_summary(char* buf, uint64_t buf_size) {
    std::map<std::string, std::pair<uint64_t, double>> agg;
    for (auto& r : g_prof) {
        if (!r.closed) continue;
        if (hipEventSynchronize(r.b) != hipSuccess) continue;
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, r.a, r.b) != hipSuccess) continue;
        auto& e = agg[r.name];
        e.first += 1;
        e.second += ms;
    }
    std::string out;
    char line[160];
    for (auto& kv : agg) {
        snprintf(line, sizeof(line), "%s,%llu,%.6f\n", kv.first.c_str(), (unsigned long long)kv.second.first,
                 kv.second.second);
        out += line;
    }
    if (buf && buf_size > 0) {
        const size_t n = out.size() < buf_size - 1 ? out.size() : buf_size - 1;
        memcpy(buf, out.data(), n);
        buf[n] = 0;
    }
    return (int64_t)out.size() + 1;
}

// ---------------------------------------------------------------------------------------------
// minimal flat JSON object reader (strings, numbers, booleans; nested values are skipped)
// ---------------------------------------------------------------------------------------------
namespace {

struct JsonVal {
    bool is_num = false;
    double num = 0.0;
    std::string str;
};
typedef std::map<std::string, JsonVal> JsonObj;

struct JsonParser {
    const char* p;
    bool ok = true;
    void ws() { while (*p == ' ' || *p == '\n' || *p == '\t' || *p == '\r') ++p; }
    std::string string_() {
        std::string s;
        if (*p != '"') { ok = false; return s; }
        ++p;
        while (*p && *p != '"') {
            if (*p == '\\' && p[1]) ++p;
            s.push_back(*p++);
        }
        if (*p == '"') ++p; else ok = false;
        return s;
    }
    void skip_nested(char open, char close) {
        int depth = 0;
        do {
            if (*p == '"') { string_(); continue; }
            if (*p == open) ++depth;
            if (*p == close) --depth;
            if (!*p) { ok = false; return; }
            ++p;
        } while (depth > 0);
    }
    bool parse(JsonObj& out) {
        ws();
        if (*p != '{') return false;
        ++p;
        while (ok) {
            ws();
            if (*p == '}') { ++p; break; }
            std::string key = string_();
            ws();
            if (*p != ':') return false;
            ++p;
            ws();
            JsonVal v;
            if (*p == '"') {
                v.str = string_();
            } else if (*p == '{') {
                skip_nested('{', '}');
            } else if (*p == '[') {
                skip_nested('[', ']');
            } else if (!strncmp(p, "true", 4)) {
                v.is_num = true; v.num = 1; v.str = "true"; p += 4;
            } else if (!strncmp(p, "false", 5)) {
                v.is_num = true; v.num = 0; v.str = "false"; p += 5;
            } else if (!strncmp(p, "null", 4)) {
                p += 4;
            } else {
                char* end = nullptr;
                v.num = strtod(p, &end);
                if (end == p) return false;
                v.is_num = true;
                p = end;
            }
            out[key] = v;
            ws();
            if (*p == ',') ++p;
        }
        return ok;
    }
};

bool json_parse(const char* text, JsonObj& out) {
    if (!text) return false;
    JsonParser jp{text};
    return jp.parse(out);
}
double json_num(const JsonObj& o, const char* key, double dflt) {
    auto it = o.find(key);
    return (it != o.end() && it->second.is_num) ? it->second.num : dflt;
}
std::string json_str(const JsonObj& o, const char* key, const char* dflt) {
    auto it = o.find(key);
    return (it != o.end() && !it->second.is_num) ? it->second.str : std::string(dflt);
}
bool json_has(const JsonObj& o, const char* key) { return o.find(key) != o.end(); }

// PCG32 (O'Neill; the generator tcnn seeds its parameter init with)
struct Pcg32 {
    uint64_t state = 0x853c49e6748fea9bULL, inc = 0xda3e39cb94b95bdbULL;
    explicit Pcg32(uint64_t seed) {
        state = 0u;
        inc = (1ULL << 1u) | 1u;
        next_uint();
        state += seed;
        next_uint();
    }
    uint32_t next_uint() {
        uint64_t old = state;
        state = old * 6364136223846793005ULL + inc;
        uint32_t xorshifted = (uint32_t)(((old >> 18u) ^ old) >> 27u);
        uint32_t rot = (uint32_t)(old >> 59u);
        return (xorshifted >> rot) | (xorshifted << ((~rot + 1u) & 31));
    }
    float next_float() {  // [0,1)
        union { uint32_t u; float f; } x;
        x.u = (next_uint() >> 9) | 0x3f800000u;
        return x.f - 1.0f;
    }
};

int parse_activation(const std::string& s, int* out) {
    if (s == "None" || s == "none") { *out = NVO_ACT_NONE; return NVO_OK; }
    if (s == "ReLU" || s == "relu") { *out = NVO_ACT_RELU; return NVO_OK; }
    if (s == "Sigmoid" || s == "sigmoid") { *out = NVO_ACT_SIGMOID; return NVO_OK; }
    nvo_set_error("activation '%s' is not supported (None, ReLU, Sigmoid)", s.c_str());
    return NVO_ERR_UNSUPPORTED;
}

}  // namespace

// ---------------------------------------------------------------------------------------------
// module objects
// ---------------------------------------------------------------------------------------------
struct nvo_module_s {
    uint32_t n_in = 0, n_out = 0, n_out_padded = 0;
    uint64_t n_params = 0;
    virtual ~nvo_module_s() {}
    virtual uint64_t ctx_bytes(uint32_t batch) const = 0;
    virtual int init_params(Pcg32& rng, float* out) const = 0;
    virtual int fwd(hipStream_t s, uint32_t B, const float* in, const void* params, void* out,
                    void* ctx) = 0;
    virtual int bwd(hipStream_t s, uint32_t B, const float* in, const void* params, const void* out,
                    const void* dout, void* ctx, float* din, float* dparams) = 0;
    virtual int set_option(const char* key, int64_t value) {
        nvo_set_error("unknown option '%s'", key);
        return NVO_ERR_INVALID;
    }
    // device ranges bwd() clears before it accumulates into them, for a given dL_dparams (nvo_bwd_zero_ranges); with
    // option "external_zero" the caller clears them instead (one launch for all networks of a training step)
    virtual int zero_ranges(float* /*dparams*/, NvoZeroRanges* /*out*/) {
        nvo_set_error("this module's backward has no externalisable zeroing");
        return NVO_ERR_UNSUPPORTED;
    }
    // parameter range [first, first + n) of this module whose Adam step the parameter backward can take over
    // (nvo_fused_adam_range / nvo_set_fused_adam); n = 0: none
    virtual int fused_adam_range(uint64_t* first, uint64_t* n) {
        *first = *n = 0;
        return NVO_OK;
    }
    // a: buffers addressed from THIS module's first parameter; NULL switches the fused step off
    virtual int set_fused_adam(const nvo_fused_adam_args* /*a*/, uint64_t /*param_offset*/) {
        nvo_set_error("this module's backward cannot take the optimiser step");
        return NVO_ERR_UNSUPPORTED;
    }
};

namespace {

struct GridModule : nvo_module_s {
    NvoGridLevels g;
    NvoGridSlices slices;
    NvoGridStream stream_bins;
    NvoGridInputScratch input_scratch;  // per-level partials of the input backward (allocated at first use)
    int bwd_mode = 1;  // 0 global atomics (the readable reference form), 1 LDS slice owner (default), 3 streamed pair records + slice owner
    bool soa_out = false;  // standalone Encoding: [B][L*F] rows (tcnn API); inside NWIE: SoA
    // option "bf16": the encoded features and their gradient are bfloat16 instead of fp16 (the format of the network
    // behind the encoding).  The table itself stays fp16, interpolation and gradient accumulation stay fp32.
    bool bf16 = false;
    int dy_fmt() const { return bf16 ? NVO_DY_BF16 : NVO_DY_HALF; }

    ~GridModule() override {
        nvo_grid_slices_destroy(&slices);
        nvo_grid_stream_destroy(&stream_bins);
        nvo_scratch_destroy(&input_scratch);
    }
    bool external_zero = false;
    const uint32_t* n_live = nullptr;  // option "n_live_ptr": device count of the rows a FORWARD evaluates (0 = all)
    int grid_zero_ranges(float* dparams, NvoZeroRanges* out) {
        int rc = ensure_slices();
        if (rc) return rc;
        if (bwd_mode == 1) {
            nvo_grid_slices_zero_ranges(g, &slices, dparams, out);
            return NVO_OK;
        }
        if (bwd_mode == 3 && nvo_grid_stream_zero_ranges(g, &stream_bins, dparams, out)) return NVO_OK;
        nvo_set_error("grid_bwd_mode %d (this layout) zeroes data-dependent ranges: external_zero is unavailable", bwd_mode);
        return NVO_ERR_UNSUPPORTED;
    }
    int zero_ranges(float* dparams, NvoZeroRanges* out) override { return grid_zero_ranges(dparams, out); }
    int set_external_zero(bool on) {
        if (on) {  // only where grid_zero_ranges can describe the zeroing
            NvoZeroRanges probe;
            if (int rc = grid_zero_ranges(nullptr, &probe)) return rc;
        }
        external_zero = on;
        slices.external_zero = on;
        stream_bins.external_zero = on;
        stream_bins.owner.external_zero = on;
        return NVO_OK;
    }
    int bwd_params(hipStream_t s, uint32_t B, const float* in, const void* dout, bool soa, float* dparams) {
        int rc = ensure_slices();
        if (rc) return rc;
        if (bwd_mode == 3) return nvo_grid_bwd_stream_launch(g, &stream_bins, s, B, in, dout, dy_fmt(), soa, dparams);
        return nvo_grid_bwd_launch(g, &slices, s, B, in, dout, dy_fmt(), soa, dparams, bwd_mode);
    }

    static int create(uint32_t n_input_dims, const JsonObj& cfg, std::unique_ptr<GridModule>* out) {
        NVO_REQUIRE(n_input_dims == 3, "HashGrid: only 3 input dims are supported (got %u)", n_input_dims);
        const std::string type = json_str(cfg, "type", "Hash");
        NVO_REQUIRE(type == "Hash", "grid type '%s' unsupported (Hash only)", type.c_str());
        const std::string interp = json_str(cfg, "interpolation", "Linear");
        NVO_REQUIRE(interp == "Linear", "interpolation '%s' unsupported (Linear only)", interp.c_str());
        const uint32_t n_levels = (uint32_t)json_num(cfg, "n_levels", 16);
        const uint32_t n_feat = (uint32_t)json_num(cfg, "n_features_per_level", 2);
        const uint32_t log2_t = (uint32_t)json_num(cfg, "log2_hashmap_size", 19);
        const uint32_t base = (uint32_t)json_num(cfg, "base_resolution", 16);
        const float pls = (float)json_num(cfg, "per_level_scale", 2.0);
        NVO_REQUIRE(n_levels >= 1 && n_levels <= NVO_MAX_LEVELS, "n_levels %u out of range", n_levels);
        NVO_REQUIRE(n_feat == 2, "n_features_per_level must be 2 (got %u)", n_feat);
        NVO_REQUIRE(log2_t >= 4 && log2_t <= 24, "log2_hashmap_size %u out of range", log2_t);
        std::unique_ptr<GridModule> m(new GridModule());
        const uint32_t entries = nvo_grid_levels_init(&m->g, n_levels, n_feat, log2_t, base, pls);
        m->n_in = 3;
        m->n_out = m->n_out_padded = n_levels * n_feat;
        m->n_params = (uint64_t)entries * n_feat;
        // The slice table of the LDS backward is a (tiny) device allocation: made lazily at the
        // first backward so that modules can be created / described on a host without a GPU.
        const char* env = getenv("NVO_GRID_BWD_MODE");
        if (env) m->bwd_mode = atoi(env);
        *out = std::move(m);
        return NVO_OK;
    }
    int fused_adam_range(uint64_t* first, uint64_t* n) override {
        *first = *n = 0;
        if (bwd_mode != 3) return NVO_OK;
        if (int rc = ensure_slices()) return rc;
        uint64_t f = 0, c = 0;
        nvo_grid_stream_adam_range(g, &stream_bins, &f, &c);
        *first = 2 * f;  // (entries -> parameters)
        *n = 2 * c;
        return NVO_OK;
    }
    int set_fused_adam(const nvo_fused_adam_args* a, uint64_t off) override {
        if (!a) {
            stream_bins.adam = NvoGridAdam{};
            return NVO_OK;
        }
        uint64_t first = 0, n = 0;
        if (int rc = fused_adam_range(&first, &n)) return rc;
        NVO_REQUIRE(n > 0, "set_fused_adam: this grid configuration has no range the backward could step (grid_bwd_mode 3, "
                           "32-bit tile-local accumulators)");
        NVO_REQUIRE(a->params && a->params_half && a->exp_avg && a->exp_avg_sq, "set_fused_adam: NULL buffer");
        NVO_REQUIRE(((((uintptr_t)(a->params + off)) | ((uintptr_t)(a->exp_avg + off)) | ((uintptr_t)(a->exp_avg_sq + off))) & 15u) == 0 &&
                        (((uintptr_t)a->params_half + 2 * off) & 7u) == 0,
                    "set_fused_adam: the encoding's parameters must start 16-byte aligned in every buffer");
        NvoGridAdam& d = stream_bins.adam;
        d.params = a->params + off;
        d.params_half = (char*)a->params_half + 2 * off;
        d.exp_avg = a->exp_avg + off;
        d.exp_avg_sq = a->exp_avg_sq + off;
        d.hyper_dev = a->hyper_dev;
        d.bias_dev = a->bias_dev;
        d.loss_scale_dev = a->loss_scale_dev;
        d.skip_flag = a->skip_flag;
        d.lr = a->lr;
        d.grad_scale = a->grad_scale;
        d.beta1 = a->beta1;
        d.beta2 = a->beta2;
        d.eps = a->eps;
        if (!a->bias_dev) {  // as nvo_adam_step / nvo_adam_step_groups price a host-side step count
            NVO_REQUIRE(a->step >= 1, "set_fused_adam: step counts from 1 (or pass bias_dev)");
            d.bias1 = 1.f - powf(a->beta1, (float)a->step);
            d.bias2_sqrt = sqrtf(1.f - powf(a->beta2, (float)a->step));
        }
        d.ema = a->ema ? a->ema + off : nullptr;
        d.ema_half = a->ema_half ? (char*)a->ema_half + 2 * off : nullptr;
        d.ema_decay = a->ema_decay;
        d.ema_step_dev = a->ema_step_dev;
        NVO_REQUIRE(!d.ema || (d.ema_step_dev && a->ema_decay >= 0.f && a->ema_decay < 1.f && (((uintptr_t)d.ema) & 15u) == 0 &&
                               (((uintptr_t)d.ema_half) & 7u) == 0),
                    "set_fused_adam: bad weight-average arguments");
        return NVO_OK;
    }
    int ensure_slices() {
        // 32-bit accumulators: half the slices per level -> fewer, longer items are the measured optimum
        if (bwd_mode == 1 && slices.n_slices == 0)
            // (32-bit accumulators + run-merging scan: 128-200 items measured best on the proposal grids -- 66.8 us per
            // launch against 79.6 us for 256-448 and 89.5 us for <= 100)
            return nvo_grid_slices_create(g, &slices, 0xFFFFFFFFu,
                                          slices.acc_bits == 32 ? (slices.runs ? 160u : 512u) : 1024u);
        if (bwd_mode == 3 && !stream_bins.created) return nvo_grid_stream_create(g, &stream_bins);
        return NVO_OK;
    }
    // option "prepare_input_gradients" (tcnn's forward flag of the same name): the forward also stores
    // d(out)/d(cell coordinate) in ctx ([L][3][B] half2) and the backward w.r.t. the input streams it instead of
    // gathering the 8 corners of every (sample, level) again.  The ctx of a forward run with the option off (or
    // with another batch size) is never read as dy/dx: the matching backward then takes the gather form.
    int prepare_input_gradients = 0;
    bool fwd_runs = false;  // option "grid_fwd_runs": the forward walks runs of four consecutive samples (k_grid_fwd_runs)
    int fwd_small_form = -1;  // option "grid_fwd_small_form": form of the small-grid forward (nvo_grid_fwd_launch), -1 = default
    bool dydx_valid = false;
    uint32_t dydx_batch = 0;
    uint64_t dydx_bytes(uint32_t B) const { return nvo_round_up((uint64_t)g.n_levels * 3 * B * 4, 256); }
    // shared by the stand-alone encoding and NetworkWithInputEncoding (dydx: where in ITS ctx the block lives)
    int fwd_encode(hipStream_t s, uint32_t B, const float* in, const void* table, void* out, bool soa, void* dydx) {
        dydx_valid = dydx != nullptr;
        dydx_batch = B;
        return nvo_grid_fwd_launch(g, s, B, in, table, out, soa, nullptr, dydx, bf16, n_live, fwd_runs, fwd_small_form);
    }
    int bwd_input(hipStream_t s, uint32_t B, const float* in, const void* table, const void* dout, bool soa,
                  float* din, const void* dydx) {
        if (dydx && dydx_valid && dydx_batch == B)
            return nvo_grid_bwd_input_dydx_launch(g, s, B, dydx, dout, dy_fmt(), soa, din, true);
        return nvo_grid_bwd_input_launch(g, s, B, in, table, dout, dy_fmt(), soa, din, true, &input_scratch);
    }
    uint64_t ctx_bytes(uint32_t B) const override { return 16 + (prepare_input_gradients ? dydx_bytes(B) : 0); }
    int init_params(Pcg32& rng, float* out) const override {
        for (uint64_t i = 0; i < n_params; ++i) out[i] = rng.next_float() * 2e-4f - 1e-4f;
        return NVO_OK;
    }
    int fwd(hipStream_t s, uint32_t B, const float* in, const void* params, void* out,
            void* ctx) override {
        void* dydx = (prepare_input_gradients && ctx) ? (char*)ctx + 16 : nullptr;
        return fwd_encode(s, B, in, params, out, soa_out, dydx);
    }
    int bwd(hipStream_t s, uint32_t B, const float* in, const void* params, const void*,
            const void* dout, void* ctx, float* din, float* dparams) override {
        if (dparams) {
            int rc = bwd_params(s, B, in, dout, soa_out, dparams);
            if (rc) return rc;
        }
        if (din) {
            const void* dydx = (prepare_input_gradients && ctx) ? (const char*)ctx + 16 : nullptr;
            int rc = bwd_input(s, B, in, params, dout, soa_out, din, dydx);
            if (rc) return rc;
        }
        return NVO_OK;
    }
    int set_option(const char* key, int64_t value) override {
        if (!strcmp(key, "grid_bwd_mode")) {
            NVO_REQUIRE(value == 0 || value == 1 || value == 3, "grid_bwd_mode: 0 (global atomics), 1 (slice owner) or 3 (streamed records)");
            bwd_mode = (int)value;
            return NVO_OK;
        }
        if (!strcmp(key, "external_zero")) return set_external_zero(value != 0);
        if (!strcmp(key, "grid_fwd_runs")) {
            fwd_runs = value != 0;
            return NVO_OK;
        }
        if (!strcmp(key, "grid_fwd_small_form")) {
            NVO_REQUIRE(value >= -1 && value <= 4, "grid_fwd_small_form: -1 (default) or 0..4");
            fwd_small_form = (int)value;
            return NVO_OK;
        }
        if (!strcmp(key, "grid_compact_live")) {  // slice-owner items scan only the samples with a non-zero gradient
            slices.compact_live = value != 0;
            return NVO_OK;
        }
        if (!strcmp(key, "bf16")) { bf16 = value != 0; return NVO_OK; }
        if (!strcmp(key, "n_live_ptr")) { n_live = reinterpret_cast<const uint32_t*>((uintptr_t)value); return NVO_OK; }
        if (!strcmp(key, "nonfinite_flag_ptr")) {  // device address of the overflow flag the backward raises (0 = none)
            slices.nf_flag = stream_bins.owner.nf_flag = reinterpret_cast<uint32_t*>((uintptr_t)value);
            return NVO_OK;
        }
        if (!strcmp(key, "deterministic")) {  // bitwise reproducible parameter gradient (see NvoGridSlices::deterministic)
            nvo_grid_slices_destroy(&slices);
            nvo_grid_stream_destroy(&stream_bins);
            slices.deterministic = stream_bins.deterministic = stream_bins.owner.deterministic = value != 0;
            return NVO_OK;
        }
        if (!strcmp(key, "prepare_input_gradients")) {  // changes ctx_bytes(): set before the ctx scratch is sized
            prepare_input_gradients = value != 0;
            dydx_valid = false;
            return NVO_OK;
        }
        if (!strcmp(key, "grid_stream_tile")) { stream_bins.tile = (uint32_t)value; return NVO_OK; }
        if (!strcmp(key, "grid_acc_bits")) {  // accumulators of the slice-owner items: 64 (default) | 32
            NVO_REQUIRE(value == 32 || value == 64, "grid_acc_bits must be 32 or 64");
            nvo_grid_slices_destroy(&slices);
            nvo_grid_stream_destroy(&stream_bins);
            slices.acc_bits = (uint32_t)value;
            stream_bins.owner.acc_bits = (uint32_t)value;
            return NVO_OK;
        }
        if (!strcmp(key, "grid_bwd_batch")) {  // expected batch size of the backward launches (0 = unknown)
            nvo_grid_slices_destroy(&slices);
            nvo_grid_stream_destroy(&stream_bins);
            slices.batch_hint = stream_bins.owner.batch_hint = (uint32_t)value;
            return NVO_OK;
        }
        if (!strcmp(key, "grid_bwd_dense_share")) {  // chunks of a dense slice relative to the even one-round split, percent
            NVO_REQUIRE(value >= 25 && value <= 400, "grid_bwd_dense_share: 25..400 (percent)");
            nvo_grid_slices_destroy(&slices);
            nvo_grid_stream_destroy(&stream_bins);
            slices.dense_share_pct = stream_bins.owner.dense_share_pct = (uint32_t)value;
            return NVO_OK;
        }
        if (!strcmp(key, "grid_bwd_runs")) {  // slice-owner items of dense levels: run-merging scan (on rebuild)
            nvo_grid_slices_destroy(&slices);
            nvo_grid_stream_destroy(&stream_bins);
            slices.runs = stream_bins.owner.runs = value != 0;
            return NVO_OK;
        }
        if (!strcmp(key, "grid_stream_overlap")) {  // 0: slice-owner levels and record pipeline back to back
            nvo_grid_stream_destroy(&stream_bins);  // (the owner's slice size depends on it)
            stream_bins.overlap = value != 0;
            return NVO_OK;
        }
        if (!strcmp(key, "grid_stream_acc_bits")) {  // (the record pass has one form left: packed 2 x 32-bit sums)
            NVO_REQUIRE(value == 32, "grid_stream_acc_bits: only the packed 32-bit record pass exists (got %lld)", (long long)value);
            return NVO_OK;
        }
        if (!strcmp(key, "grid_stream_owner_slices")) {  // takes effect when the tables are (re)built
            nvo_grid_stream_destroy(&stream_bins);
            stream_bins.owner_max_slices = (uint32_t)value;
            return NVO_OK;
        }
        return nvo_module_s::set_option(key, value);
    }
};

struct ShModule : nvo_module_s {
    uint32_t degree = 4;
    bool bf16 = false;  // option "bf16": bfloat16 output
    int set_option(const char* key, int64_t value) override {
        if (!strcmp(key, "bf16")) { bf16 = value != 0; return NVO_OK; }
        return nvo_module_s::set_option(key, value);
    }
    static int create(uint32_t n_input_dims, const JsonObj& cfg, std::unique_ptr<ShModule>* out) {
        NVO_REQUIRE(n_input_dims == 3, "SphericalHarmonics needs 3 input dims (got %u)", n_input_dims);
        std::unique_ptr<ShModule> m(new ShModule());
        m->degree = (uint32_t)json_num(cfg, "degree", 4);
        NVO_REQUIRE(m->degree >= 1 && m->degree <= 4, "SphericalHarmonics degree %u not in 1..4", m->degree);
        m->n_in = 3;
        m->n_out = m->n_out_padded = m->degree * m->degree;
        m->n_params = 0;
        *out = std::move(m);
        return NVO_OK;
    }
    uint64_t ctx_bytes(uint32_t) const override { return 16; }
    int init_params(Pcg32&, float*) const override { return NVO_OK; }
    int fwd(hipStream_t s, uint32_t B, const float* in, const void*, void* out, void*) override {
        return nvo_sh_fwd_launch(s, B, degree, in, out, n_out_padded, n_out_padded, bf16);
    }
    int bwd(hipStream_t s, uint32_t B, const float* in, const void*, const void*, const void* dout,
            void*, float* din, float*) override {
        NVO_REQUIRE(!(din && bf16), "SphericalHarmonics: the input gradient of the bf16 form is not built");
        if (din) return nvo_sh_bwd_input_launch(s, B, degree, in, dout, n_out_padded, din);
        return NVO_OK;
    }
};

struct MlpModule : nvo_module_s {
    int in_pad = 16, width = 64, n_hidden = 1, out_pad = 16;
    int act = NVO_ACT_RELU, out_act = NVO_ACT_NONE;
    // option "bf16": weights, hidden activations, output and dL/doutput are bfloat16 and the layers run on
    // v_mfma_f32_16x16x16_bf16 (mlp_bf16.hip) instead of fp16 / v_mfma_f32_16x16x16_f16; accumulation is fp32 in both
    int bf16 = 0;
    // option "deterministic": the weight gradient is summed over the workgroups in a fixed order (block totals stored
    // to this scratch + a reduce launch) instead of float atomics
    bool deterministic = false;
    // option "nonfinite_flag_ptr": device uint32 the backward ORs with 1 when a weight-gradient total is not finite
    uint32_t* nf_flag = nullptr;
    // options "dw_replicas_ptr" / "dw_replicas": caller-owned zeroed copies of the weight-gradient buffer the backward's
    // workgroups spread their adds over (NvoMlpArgsT::dw_replicas; the caller folds them with nvo_fold_replicas)
    float* dw_replicas = nullptr;
    uint32_t dw_n_replicas = 0;
    NvoScratch dw_scratch;
    ~MlpModule() override { nvo_scratch_destroy(&dw_scratch); }
    int det_partials(hipStream_t s, uint32_t B, NvoMlpArgs* a) {
        if (!deterministic || !a->dweights) return NVO_OK;
        const size_t bytes = sizeof(float) * (size_t)nvo_mlp_bwd_blocks(in_pad, width, n_hidden, B) *
                             nvo_mlp_n_weights(in_pad, width, n_hidden, out_pad);
        if (int rc = nvo_scratch_reserve(&dw_scratch, bytes, s, "mlp dW block totals")) return rc;
        a->dw_partial = static_cast<float*>(dw_scratch.ptr);
        return NVO_OK;
    }
    int set_option(const char* key, int64_t value) override {
        if (!strcmp(key, "bf16")) { bf16 = value != 0; return NVO_OK; }
        if (!strcmp(key, "external_zero")) { external_zero = value != 0; return NVO_OK; }
        if (!strcmp(key, "deterministic")) { deterministic = value != 0; return NVO_OK; }
        if (!strcmp(key, "nonfinite_flag_ptr")) { nf_flag = reinterpret_cast<uint32_t*>((uintptr_t)value); return NVO_OK; }
        if (!strcmp(key, "dw_replicas_ptr")) { dw_replicas = reinterpret_cast<float*>((uintptr_t)value); return NVO_OK; }
        if (!strcmp(key, "dw_replicas")) {
            NVO_REQUIRE(value >= 0 && value <= 63, "dw_replicas: 0..63 copies");
            dw_n_replicas = (uint32_t)value;
            return NVO_OK;
        }
        return nvo_module_s::set_option(key, value);
    }

    static int create(uint32_t n_input_dims, uint32_t n_output_dims, const JsonObj& cfg,
                      std::unique_ptr<MlpModule>* out) {
        const std::string otype = json_str(cfg, "otype", "FullyFusedMLP");
        NVO_REQUIRE(otype == "FullyFusedMLP" || otype == "CutlassMLP",
                    "network otype '%s' unsupported", otype.c_str());
        std::unique_ptr<MlpModule> m(new MlpModule());
        m->width = (int)json_num(cfg, "n_neurons", 64);
        m->n_hidden = (int)json_num(cfg, "n_hidden_layers", 1);
        int rc = parse_activation(json_str(cfg, "activation", "ReLU"), &m->act);
        if (rc) return rc;
        rc = parse_activation(json_str(cfg, "output_activation", "None"), &m->out_act);
        if (rc) return rc;
        m->in_pad = (int)nvo_round_up(n_input_dims, 16);
        m->out_pad = (int)nvo_round_up(n_output_dims, 16);
        m->n_in = n_input_dims;
        m->n_out = n_output_dims;
        m->n_out_padded = (uint32_t)m->out_pad;
        if (!nvo_mlp_shape_supported(m->in_pad, m->width, m->n_hidden, m->out_pad)) {
            nvo_set_error("FullyFusedMLP shape (in_pad=%d, n_neurons=%d, n_hidden_layers=%d, out_pad=%d) "
                          "has no gfx950 kernel instance", m->in_pad, m->width, m->n_hidden, m->out_pad);
            return NVO_ERR_UNSUPPORTED;
        }
        m->n_params = (uint64_t)m->width * m->in_pad + (uint64_t)(m->n_hidden - 1) * m->width * m->width +
                      (uint64_t)m->out_pad * m->width;
        *out = std::move(m);
        return NVO_OK;
    }
    uint64_t hidden_bytes(uint32_t B) const { return (uint64_t)n_hidden * B * width * sizeof(_Float16); }
    uint64_t ctx_bytes(uint32_t B) const override { return nvo_round_up(hidden_bytes(B), 256) + 256; }
    int init_params(Pcg32& rng, float* out) const override {
        // Xavier uniform per weight matrix (tcnn FullyFusedMLP::initialize_params)
        uint64_t o = 0;
        auto fill = [&](int fan_out, int fan_in) {
            const float scale = sqrtf(6.0f / (float)(fan_in + fan_out));
            for (int i = 0; i < fan_out * fan_in; ++i) out[o++] = (rng.next_float() * 2.f - 1.f) * scale;
        };
        fill(width, in_pad);
        for (int l = 0; l < n_hidden - 1; ++l) fill(width, width);
        fill(out_pad, width);
        return NVO_OK;
    }
    NvoMlpArgs make_args(uint32_t B, const void* in, int in_mode, uint32_t n_in_true,
                         const void* params, void* out, void* ctx) const {
        NvoMlpArgs a;
        memset(&a, 0, sizeof(a));
        a.batch = B;
        a.n_in = n_in_true;
        a.in_mode = in_mode;
        a.input = in;
        a.weights = (const _Float16*)params;
        a.output = (_Float16*)out;
        a.hidden = (_Float16*)ctx;
        a.act = act;
        a.out_act = out_act;
        a.bf16 = bf16;
        a.nf_flag = nf_flag;
        if (dw_replicas && dw_n_replicas && !deterministic) {
            a.dw_replicas = dw_replicas;
            a.dw_n_replicas = dw_n_replicas;
        }
        return a;
    }
    int fwd(hipStream_t s, uint32_t B, const float* in, const void* params, void* out,
            void* ctx) override {
        NvoMlpArgs a = make_args(B, in, NVO_IO_F32_ROWS, n_in, params, out, ctx);
        return nvo_mlp_fwd_launch(in_pad, width, n_hidden, out_pad, a, s);
    }
    int bwd(hipStream_t s, uint32_t B, const float* in, const void* params, const void* out,
            const void* dout, void* ctx, float* din, float* dparams) override {
        NVO_REQUIRE(ctx != nullptr, "Network.bwd needs the ctx of the matching fwd");
        NvoMlpArgs a = make_args(B, in, NVO_IO_F32_ROWS, n_in, params, (void*)out, ctx);
        a.doutput = (const _Float16*)dout;
        a.dinput = din;
        a.din_mode = NVO_IO_F32_ROWS;
        a.dweights = dparams;
        if (dparams && !external_zero)
            if (int rc = nvo_zero_async(dparams, sizeof(float) * n_params, s)) return rc;
        if (int rc = det_partials(s, B, &a)) return rc;
        return nvo_mlp_bwd_launch(in_pad, width, n_hidden, out_pad, a, s);
    }
    bool external_zero = false;
    int zero_ranges(float* dparams, NvoZeroRanges* out) override {
        out->push_back({dparams, sizeof(float) * n_params});
        return NVO_OK;
    }
};

// tcnn NetworkWithInputEncoding: params = [network | encoding]; the encoded features stay
// level-major fp16 in ctx between the two kernels.
struct NwieModule : nvo_module_s {
    std::unique_ptr<GridModule> enc;
    std::unique_ptr<MlpModule> net;
    int compact_out = 0;  // option "compact_output": output / dL_doutput are [B] halfs (column 0 only)
    int recompute_hidden = 0;  // option "recompute_hidden": the forward does not store the hidden layer
    // options "bwd_tile_live_ptr" / "bwd_tile_live_bits": NvoMlpArgsT::tile_live of the backwards that follow (the caller
    // clears the pointer when its dL/doutput no longer comes with such bytes)
    const uint8_t* bwd_tile_live = nullptr;
    uint32_t bwd_tile_live_bits = 0xffu;
    const float* bwd_tile_live_count = nullptr;  // option "bwd_tile_live_count_ptr"
    // option "fuse_encoding": the forward evaluates the hash grid inside the MLP kernel's operand load
    // (NVO_IO_GRID_FUSED) -- one launch, no feature round trip between two kernels.  Meant for grids whose tables fit
    // every XCD's L2 (the proposal networks: 1.5 MB): the stand-alone k_grid_fwd keeps a level's table on ONE XCD,
    // which a fused kernel cannot.  Bit-identical outputs.
    int fuse_encoding = 0;
    NvoGridLevels* d_levels = nullptr;  // device copy of enc->g for the fused kernel

    int fused_adam_range(uint64_t* first, uint64_t* n) override {
        if (int rc = enc->fused_adam_range(first, n)) return rc;
        if (*n) *first += net->n_params;  // params = [network | encoding]
        return NVO_OK;
    }
    int set_fused_adam(const nvo_fused_adam_args* a, uint64_t off) override {
        return enc->set_fused_adam(a, off + net->n_params);
    }
    uint64_t enc_bytes(uint32_t B) const { return nvo_round_up((uint64_t)enc->g.n_levels * B * 4, 256); }
    uint64_t ctx_bytes(uint32_t B) const override {
        // [encoded SoA][d_encoded SoA][mlp hidden][dy/dx of the encoding (option prepare_input_gradients)]
        return dydx_offset(B) + (enc->prepare_input_gradients ? enc->dydx_bytes(B) : 0);
    }
    uint64_t dydx_offset(uint32_t B) const { return nvo_round_up(2 * enc_bytes(B) + net->ctx_bytes(B), 256); }
    int init_params(Pcg32& rng, float* out) const override {
        int rc = net->init_params(rng, out);
        if (rc) return rc;
        return enc->init_params(rng, out + net->n_params);
    }
    int fwd(hipStream_t s, uint32_t B, const float* in, const void* params, void* out,
            void* ctx) override {
        NVO_REQUIRE(ctx != nullptr, "NetworkWithInputEncoding.fwd needs ctx scratch (also for inference)");
        char* c = (char*)ctx;
        void* encoded = c;
        void* hidden = c + 2 * enc_bytes(B);
        const _Float16* p = (const _Float16*)params;
        void* dydx = enc->prepare_input_gradients ? c + dydx_offset(B) : nullptr;
        if (fuse_encoding && !dydx && !enc->n_live && net->n_hidden == 1 && net->in_pad <= 32 && net->out_pad == 16) {
            if (!d_levels) {  // (first call = an eager warm-up step, never under graph capture)
                NVO_CHECK_HIP(hipMalloc((void**)&d_levels, sizeof(NvoGridLevels)));
                NVO_CHECK_HIP(hipMemcpy(d_levels, &enc->g, sizeof(NvoGridLevels), hipMemcpyHostToDevice));
            }
            NvoMlpArgs a = net->make_args(B, in, NVO_IO_GRID_FUSED, enc->n_out, p, out, hidden);
            a.compact_out = compact_out;
            if (recompute_hidden) a.hidden = nullptr;
            a.grid = d_levels;
            a.grid_table = p + net->n_params;
            a.enc_out = encoded;  // the backward (and the recomputed hidden layer) read the features from ctx
            return nvo_mlp_fwd_launch(net->in_pad, net->width, net->n_hidden, net->out_pad, a, s);
        }
        int rc = enc->fwd_encode(s, B, in, p + net->n_params, encoded, true, dydx);
        if (rc) return rc;
        NvoMlpArgs a = net->make_args(B, encoded, NVO_IO_HALF2_SOA, enc->n_out, p, out, hidden);
        a.compact_out = compact_out;
        if (recompute_hidden) a.hidden = nullptr;
        a.n_live = enc->n_live;  // (option "n_live_ptr": both kernels stop at the rows in use)
        return nvo_mlp_fwd_launch(net->in_pad, net->width, net->n_hidden, net->out_pad, a, s);
    }
    int zero_ranges(float* dparams, NvoZeroRanges* out) override {
        if (int rc = net->zero_ranges(dparams, out)) return rc;
        return enc->grid_zero_ranges(dparams ? dparams + net->n_params : nullptr, out);
    }
    hipEvent_t ev_fork = nullptr;  // nvo_bwd_fork: network backward done -> the encoding's parameter backward may start
    // per-workgroup L1 sums (and live-sample counts) of dL/d(encoded), written by the network's backward and read by the
    // encoding's 32-bit slice-owner items: NvoMlpArgsT::dx_l1_partial.  [blocks][in_pad] floats | [blocks] uint32
    NvoScratch l1_scratch;
    ~NwieModule() override {
        if (ev_fork) (void)hipEventDestroy(ev_fork);
        if (d_levels) (void)hipFree(d_levels);
        nvo_scratch_destroy(&l1_scratch);
    }
    int bwd(hipStream_t s, uint32_t B, const float* in, const void* params, const void* out,
            const void* dout, void* ctx, float* din, float* dparams) override {
        return bwd_on(s, s, B, in, params, out, dout, ctx, din, dparams);
    }
    // sp: stream of the encoding's parameter backward (== s: everything in order on one stream)
    int bwd_on(hipStream_t s, hipStream_t sp, uint32_t B, const float* in, const void* params, const void* out,
               const void* dout, void* ctx, float* din, float* dparams) {
        NVO_REQUIRE(ctx != nullptr, "NetworkWithInputEncoding.bwd needs the ctx of the matching fwd");
        char* c = (char*)ctx;
        void* encoded = c;
        void* dencoded = c + enc_bytes(B);
        void* hidden = c + 2 * enc_bytes(B);
        const _Float16* p = (const _Float16*)params;
        NvoMlpArgs a = net->make_args(B, encoded, NVO_IO_HALF2_SOA, enc->n_out, p, (void*)out, hidden);
        a.doutput = (const _Float16*)dout;
        a.compact_out = compact_out;
        a.recompute_hidden = recompute_hidden;
        a.dinput = dencoded;
        a.din_mode = NVO_IO_HALF2_SOA;
        a.dweights = dparams;
        const char* const e_dead = getenv("NVO_MLP_SKIP_DEAD");  // (A/B, tests: 0 = every tile in its turn; per launch)
        if (bwd_tile_live && !compact_out && !(e_dead && atoi(e_dead) == 0)) {  // (module option "bwd_tile_live_ptr": the caller's promise about `dout`)
            a.tile_live = bwd_tile_live;
            a.tile_live_bits = bwd_tile_live_bits;
            a.tile_live_count = bwd_tile_live_count;
        }
        if (dparams && !net->external_zero)
            if (int rc0 = nvo_zero_async(dparams, sizeof(float) * net->n_params, s)) return rc0;
        if (int rc0 = net->det_partials(s, B, &a)) return rc0;
        // the samples behind dead tiles carry no gradient: the encoding walks a LIST of the others, which this network's
        // backward writes while it walks its live tiles (NvoMlpArgsT::live_rows) -- or, where it cannot, a pass of the
        // encoding's own over `dout` (k_live_rows)
        const char* const e_rows = getenv("NVO_GRID_LIVE_ROWS");  // A/B, tests (per launch; a graph keeps its capture's)
        const NvoGridSlices* const row_owner = enc->bwd_mode == 3 ? &enc->stream_bins.owner : nullptr;
        const bool list_rows = (!e_rows || atoi(e_rows) != 0) && a.tile_live && dparams && row_owner && row_owner->d_live_n &&
                               !row_owner->deterministic && net->out_pad == 16 && recompute_hidden && (B & 15u) == 0u;
        bool rows_by_mlp = false;
        if (list_rows && nvo_mlp_bwd_lists_rows(net->in_pad, net->width, net->n_hidden, B)) {
            const char* const e_mlp = getenv("NVO_MLP_LISTS_ROWS");  // (A/B: 0 = the encoding's own pass)
            rows_by_mlp = !e_mlp || atoi(e_mlp) != 0;
        }
        if (rows_by_mlp) {
            if (int rc0 = nvo_scratch_reserve(&row_owner->live, sizeof(uint32_t) * ((size_t)B + 1), s, "grid_bwd live list")) return rc0;
            if (!row_owner->external_zero)  // (otherwise cleared by the step's zero launch: nvo_grid_slices_zero_ranges)
                if (int rc0 = nvo_zero_async(row_owner->d_live_n, sizeof(uint32_t), s)) return rc0;
            a.live_rows = static_cast<uint32_t*>(row_owner->live.ptr);
            a.live_rows_n = row_owner->d_live_n;
        }
        // the encoding's 32-bit accumulators (slice-owner items) scale by the L1 norm of dL/d(encoded): the network's
        // backward sums it while it stores those values -- no pass of its own over them
        static const bool l1_from_mlp = [] { const char* e = getenv("NVO_GRID_L1_FROM_MLP"); return !e || atoi(e) != 0; }();
        const NvoGridSlices* owner = enc->bwd_mode == 1 ? &enc->slices : enc->bwd_mode == 3 ? &enc->stream_bins.owner : nullptr;
        const bool want_l1 = l1_from_mlp && dparams && owner && owner->acc_bits == 32;
        const uint32_t l1_blocks = want_l1 ? nvo_mlp_bwd_blocks(net->in_pad, net->width, net->n_hidden, B) : 0u;
        if (want_l1) {
            const size_t bytes = sizeof(float) * (size_t)l1_blocks * (net->in_pad + 1);
            if (int rc0 = nvo_scratch_reserve(&l1_scratch, bytes, s, "grid L1 partials")) return rc0;
            a.dx_l1_partial = static_cast<float*>(l1_scratch.ptr);
            a.dx_live_partial = compact_out ? reinterpret_cast<uint32_t*>(a.dx_l1_partial + (size_t)l1_blocks * net->in_pad) : nullptr;
        }
        int rc = nvo_mlp_bwd_launch(net->in_pad, net->width, net->n_hidden, net->out_pad, a, s);
        if (rc) return rc;
        if (dparams) {
            if (sp != s) {  // fork (the first call happens in an eager warm-up step, never under graph capture)
                if (!ev_fork) NVO_CHECK_HIP(hipEventCreateWithFlags(&ev_fork, hipEventDisableTiming));
                NVO_CHECK_HIP(hipEventRecord(ev_fork, s));
                NVO_CHECK_HIP(hipStreamWaitEvent(sp, ev_fork, 0));
            }
            if (want_l1) {
                owner->ext_l1 = a.dx_l1_partial;
                owner->ext_live = a.dx_live_partial;
                static const bool live_dout = [] { const char* e = getenv("NVO_LIVE_FROM_DOUT"); return !e || atoi(e) != 0; }();  // A/B
                owner->ext_dout = (live_dout && compact_out && a.dx_live_partial) ? reinterpret_cast<const uint16_t*>(dout) : nullptr;
                owner->ext_blocks = l1_blocks;
                owner->ext_l1_stride = (uint32_t)net->in_pad;
            }
            if (rows_by_mlp) {
                row_owner->ext_list_given = true;  // (the list and its length word are in place: a.live_rows above)
            } else if (list_rows) {
                row_owner->ext_tile_live = a.tile_live;
                row_owner->ext_tile_bits = a.tile_live_bits;
                row_owner->ext_rows = dout;
                row_owner->ext_tile_count = bwd_tile_live_count;
            }
            rc = enc->bwd_params(sp, B, in, dencoded, true, dparams + net->n_params);
            if (list_rows) {
                row_owner->ext_list_given = false;
                row_owner->ext_tile_live = nullptr;
                row_owner->ext_rows = nullptr;
                row_owner->ext_tile_count = nullptr;
                row_owner->ext_tile_bits = 0u;
            }
            if (want_l1) {
                owner->ext_l1 = nullptr;
                owner->ext_live = nullptr;
                owner->ext_dout = nullptr;
                owner->ext_blocks = owner->ext_l1_stride = 0u;
            }
            if (rc) return rc;
        }
        if (din) {
            const void* dydx = enc->prepare_input_gradients ? c + dydx_offset(B) : nullptr;
            rc = enc->bwd_input(s, B, in, p + net->n_params, dencoded, true, din, dydx);
            if (rc) return rc;
        }
        return NVO_OK;
    }
    int set_option(const char* key, int64_t value) override {
        if (!strcmp(key, "compact_output")) {
            compact_out = value != 0;
            return NVO_OK;
        }
        if (!strcmp(key, "bf16")) {  // network in bfloat16, encoding output / gradient in bfloat16, table fp16
            net->bf16 = value != 0;
            return enc->set_option(key, value);
        }
        if (!strcmp(key, "fuse_encoding")) {
            fuse_encoding = value != 0;
            return NVO_OK;
        }
        if (!strcmp(key, "deterministic")) {  // network weight gradient AND hash-grid gradient bitwise reproducible
            net->deterministic = value != 0;
            return enc->set_option(key, value);
        }
        if (!strcmp(key, "nonfinite_flag_ptr")) {  // leaves (encoding's parameter scatter) and the network's dW flush
            net->nf_flag = reinterpret_cast<uint32_t*>((uintptr_t)value);
            return enc->set_option(key, value);
        }
        if (!strcmp(key, "dw_replicas_ptr") || !strcmp(key, "dw_replicas")) return net->set_option(key, value);
        if (!strcmp(key, "external_zero")) {
            if (int rc = enc->set_external_zero(value != 0)) return rc;
            net->external_zero = value != 0;
            return NVO_OK;
        }
        if (!strcmp(key, "bwd_tile_live_ptr")) {  // device bytes, one per 16 samples of the NEXT backwards' dL/doutput (0 = none)
            bwd_tile_live = reinterpret_cast<const uint8_t*>((uintptr_t)value);
            return NVO_OK;
        }
        if (!strcmp(key, "bwd_tile_live_bits")) {
            bwd_tile_live_bits = (uint32_t)value;
            return NVO_OK;
        }
        if (!strcmp(key, "debug_copy_grid_live_n")) {  // (tests) length word of the encoding's live-sample list -> *value (device u32)
            const NvoGridSlices* owner = enc->bwd_mode == 1 ? &enc->slices : enc->bwd_mode == 3 ? &enc->stream_bins.owner : nullptr;
            NVO_REQUIRE(owner && owner->d_live_n && value, "debug_copy_grid_live_n: no slice-owner state");
            NVO_CHECK_HIP(hipMemcpy(reinterpret_cast<void*>((uintptr_t)value), owner->d_live_n, sizeof(uint32_t), hipMemcpyDeviceToDevice));
            return NVO_OK;
        }
        if (!strcmp(key, "bwd_tile_live_count_ptr")) {  // 64 float shards, 8 floats apart: their sum = number of live tiles (0 = none)
            bwd_tile_live_count = reinterpret_cast<const float*>((uintptr_t)value);
            return NVO_OK;
        }
        if (!strcmp(key, "recompute_hidden")) {
            NVO_REQUIRE(value == 0 || (net->n_hidden == 1 && net->act == NVO_ACT_RELU),
                        "recompute_hidden needs a single hidden layer with ReLU");
            recompute_hidden = value != 0;
            return NVO_OK;
        }
        return enc->set_option(key, value);
    }
};

int create_encoding_impl(uint32_t n_input_dims, const char* json, std::unique_ptr<nvo_module_s>* out,
                         std::unique_ptr<GridModule>* grid_out) {
    JsonObj cfg;
    NVO_REQUIRE(json_parse(json, cfg), "encoding config is not a JSON object: %s", json ? json : "(null)");
    const std::string otype = json_str(cfg, "otype", "");
    if (otype == "HashGrid" || otype == "Grid") {
        std::unique_ptr<GridModule> g;
        int rc = GridModule::create(n_input_dims, cfg, &g);
        if (rc) return rc;
        if (grid_out) *grid_out = std::move(g); else *out = std::move(g);
        return NVO_OK;
    }
    if (otype == "SphericalHarmonics") {
        NVO_REQUIRE(grid_out == nullptr, "NetworkWithInputEncoding supports HashGrid encodings only");
        std::unique_ptr<ShModule> m;
        int rc = ShModule::create(n_input_dims, cfg, &m);
        if (rc) return rc;
        *out = std::move(m);
        return NVO_OK;
    }
    nvo_set_error("encoding otype '%s' unsupported (HashGrid, SphericalHarmonics)", otype.c_str());
    return NVO_ERR_UNSUPPORTED;
}

}  // namespace

// ---------------------------------------------------------------------------------------------
// exported C-ABI
// ---------------------------------------------------------------------------------------------
extern "C" {

int nvo_create_encoding(uint32_t n_input_dims, const char* encoding_json, nvo_module_t* out) {
    NVO_REQUIRE(out != nullptr, "out is NULL");
    std::unique_ptr<nvo_module_s> m;
    int rc = create_encoding_impl(n_input_dims, encoding_json, &m, nullptr);
    if (rc) return rc;
    *out = m.release();
    return NVO_OK;
}

int nvo_create_network(uint32_t n_input_dims, uint32_t n_output_dims, const char* network_json,
                       nvo_module_t* out) {
    NVO_REQUIRE(out != nullptr, "out is NULL");
    JsonObj cfg;
    NVO_REQUIRE(json_parse(network_json, cfg), "network config is not a JSON object");
    std::unique_ptr<MlpModule> m;
    int rc = MlpModule::create(n_input_dims, n_output_dims, cfg, &m);
    if (rc) return rc;
    *out = m.release();
    return NVO_OK;
}

int nvo_create_network_with_input_encoding(uint32_t n_input_dims, uint32_t n_output_dims,
                                           const char* encoding_json, const char* network_json,
                                           nvo_module_t* out) {
    NVO_REQUIRE(out != nullptr, "out is NULL");
    std::unique_ptr<NwieModule> m(new NwieModule());
    std::unique_ptr<nvo_module_s> unused;
    int rc = create_encoding_impl(n_input_dims, encoding_json, &unused, &m->enc);
    if (rc) return rc;
    m->enc->soa_out = true;
    JsonObj cfg;
    NVO_REQUIRE(json_parse(network_json, cfg), "network config is not a JSON object");
    rc = MlpModule::create(m->enc->n_out, n_output_dims, cfg, &m->net);
    if (rc) return rc;
    m->n_in = n_input_dims;
    m->n_out = n_output_dims;
    m->n_out_padded = m->net->n_out_padded;
    m->n_params = m->net->n_params + m->enc->n_params;
    *out = m.release();
    return NVO_OK;
}

int nvo_destroy(nvo_module_t m) {
    delete m;
    return NVO_OK;
}

uint32_t nvo_n_input_dims(nvo_module_t m) { return m ? m->n_in : 0; }
uint32_t nvo_n_output_dims(nvo_module_t m) { return m ? m->n_out : 0; }
uint32_t nvo_padded_output_dims(nvo_module_t m) { return m ? m->n_out_padded : 0; }
uint64_t nvo_n_params(nvo_module_t m) { return m ? m->n_params : 0; }
uint64_t nvo_ctx_bytes(nvo_module_t m, uint32_t batch) { return m ? m->ctx_bytes(batch) : 0; }

int nvo_initial_params(nvo_module_t m, uint64_t seed, float* host_out) {
    NVO_REQUIRE(m && (host_out || m->n_params == 0), "initial_params: NULL argument");
    Pcg32 rng(seed);
    return m->init_params(rng, host_out);
}

int nvo_set_option(nvo_module_t m, const char* key, int64_t value) {
    NVO_REQUIRE(m && key, "set_option: NULL argument");
    return m->set_option(key, value);
}

int nvo_fused_adam_range(nvo_module_t m, uint64_t* first_param, uint64_t* n_params) {
    NVO_REQUIRE(m && first_param && n_params, "fused_adam_range: NULL argument");
    return m->fused_adam_range(first_param, n_params);
}

int nvo_set_fused_adam(nvo_module_t m, const nvo_fused_adam_args* args) {
    NVO_REQUIRE(m, "set_fused_adam: NULL module");
    return m->set_fused_adam(args, 0);
}

int nvo_bwd_zero_ranges(nvo_module_t m, float* dL_dparams, void** ptrs_out, uint64_t* bytes_out, uint32_t capacity) {
    NVO_REQUIRE(m && dL_dparams && ptrs_out && bytes_out, "bwd_zero_ranges: NULL argument");
    NvoZeroRanges r;
    if (int rc = m->zero_ranges(dL_dparams, &r)) return rc < 0 ? rc : -rc;
    if (r.size() > capacity) {
        nvo_set_error("bwd_zero_ranges: %zu ranges, capacity %u", r.size(), capacity);
        return NVO_ERR_INVALID < 0 ? NVO_ERR_INVALID : -NVO_ERR_INVALID;
    }
    for (size_t i = 0; i < r.size(); ++i) {
        ptrs_out[i] = r[i].first;
        bytes_out[i] = r[i].second;
    }
    return (int)r.size();
}

int nvo_fwd(nvo_module_t m, nvo_stream_t stream, uint32_t batch, const float* input,
            const void* params, void* output, void* ctx) {
    NVO_REQUIRE(m, "fwd: NULL module");
    if (batch == 0) return NVO_OK;  // an empty batch has no rows to write (its buffers may be NULL: torch's empty tensors)
    NVO_REQUIRE(output && input, "fwd: NULL argument");
    NVO_REQUIRE(params || m->n_params == 0, "fwd: params is NULL");
    NVO_REQUIRE((batch & 15u) == 0, "fwd: batch (%u) must be a multiple of 16", batch);
    return m->fwd((hipStream_t)stream, batch, input, params, output, ctx);
}

int nvo_bwd(nvo_module_t m, nvo_stream_t stream, uint32_t batch, const float* input,
            const void* params, const void* output, const void* dL_doutput, void* ctx,
            float* dL_dinput, float* dL_dparams) {
    NVO_REQUIRE(m, "bwd: NULL module");
    if (batch == 0) {  // the gradient of an empty batch: zeros (dL_dinput has no rows)
        if (dL_dparams && m->n_params) return nvo_zero_async(dL_dparams, sizeof(float) * m->n_params, (hipStream_t)stream);
        return NVO_OK;
    }
    NVO_REQUIRE(dL_doutput && input, "bwd: NULL argument");
    NVO_REQUIRE((batch & 15u) == 0, "bwd: batch (%u) must be a multiple of 16", batch);
    return m->bwd((hipStream_t)stream, batch, input, params, output, dL_doutput, ctx, dL_dinput,
                  dL_dparams);
}

int nvo_bwd_fork(nvo_module_t m, nvo_stream_t stream, nvo_stream_t params_stream, uint32_t batch,
                 const float* input, const void* params, const void* output, const void* dL_doutput, void* ctx,
                 float* dL_dinput, float* dL_dparams) {
    NVO_REQUIRE(m && dL_doutput && (input || batch == 0), "bwd_fork: NULL argument");
    NVO_REQUIRE((batch & 15u) == 0, "bwd_fork: batch (%u) must be a multiple of 16", batch);
    auto* n = dynamic_cast<NwieModule*>(m);
    if (!n || !params_stream || params_stream == stream || !dL_dparams)
        return m->bwd((hipStream_t)stream, batch, input, params, output, dL_doutput, ctx, dL_dinput, dL_dparams);
    return n->bwd_on((hipStream_t)stream, (hipStream_t)params_stream, batch, input, params, output, dL_doutput, ctx,
                     dL_dinput, dL_dparams);
}

static GridModule* as_grid(nvo_module_t m) {
    if (auto* g = dynamic_cast<GridModule*>(m)) return g;
    if (auto* n = dynamic_cast<NwieModule*>(m)) return n->enc.get();
    return nullptr;
}

int nvo_grid_describe(nvo_module_t m, uint32_t* levels_out, float* scales_out) {
    GridModule* g = as_grid(m);
    NVO_REQUIRE(g && levels_out && scales_out, "grid_describe: module has no grid encoding");
    for (uint32_t l = 0; l < g->g.n_levels; ++l) {
        levels_out[4 * l + 0] = g->g.offset[l];
        levels_out[4 * l + 1] = g->g.offset[l + 1] - g->g.offset[l];
        levels_out[4 * l + 2] = g->g.resolution[l];
        levels_out[4 * l + 3] = g->g.hashed[l];
        scales_out[l] = g->g.scale[l];
    }
    return NVO_OK;
}

int nvo_grid_indices(nvo_module_t m, nvo_stream_t stream, uint32_t batch, const float* input,
                     uint32_t* indices_out) {
    GridModule* g = as_grid(m);
    NVO_REQUIRE(g && indices_out, "grid_indices: module has no grid encoding");
    // The forward kernel needs a table and an output; allocate throw-away ones (debug path only).
    void* table = nullptr;
    void* out = nullptr;
    NVO_CHECK_HIP(hipMalloc(&table, (size_t)g->n_params * 2));
    NVO_CHECK_HIP(hipMemsetAsync(table, 0, (size_t)g->n_params * 2, (hipStream_t)stream));
    NVO_CHECK_HIP(hipMalloc(&out, (size_t)g->g.n_levels * batch * 4 + 16));
    int rc = nvo_grid_fwd_launch(g->g, (hipStream_t)stream, batch, input, table, out, true, indices_out);
    hipError_t e = hipStreamSynchronize((hipStream_t)stream);
    (void)hipFree(table);
    (void)hipFree(out);
    if (rc) return rc;
    NVO_CHECK_HIP(e);
    return NVO_OK;
}

}  // extern "C"
