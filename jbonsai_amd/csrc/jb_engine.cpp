// jb_engine.cpp -- engine-level C ABI (include/jbonsai_amd.h, part 2).
//
// Host front half (cold, per utterance O(labels)): label lines -> per-state pdfs
// (decision-tree search, multi-voice blend) -> state durations.  Mirrors
//   Engine / Condition            src/engine.rs:31-366
//   Labels::load_from_strings     src/label.rs:35-113
//   Models::{duration,stream,gv}  src/model/mod.rs:80-156
//   VoiceSet::{new,weighted}      src/model/voice_set.rs:22-95
//   DurationEstimator             src/duration.rs:20-131
//   InterporationWeight           src/model/interporation_weight.rs:48-160
// Everything from the state level down (MLPG, GV, excitation, MLSA) is handed to
// the HIP batch (jb_batch.cpp); nothing here synthesises audio.
#include "jb_host.h"
#include "jb_voice.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <map>
#include <mutex>
#include <thread>
#include <cerrno>
#include <cfloat>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <sys/mman.h>

namespace jb {

constexpr double kDB = 0.11512925464970228;        // ln(10)/20, src/constants.rs:11
constexpr double kHalfTone = 0.05776226504666211;  // ln(2)/12,  src/constants.rs:9
constexpr double kMaxLf0 = 9.903487552536127, kMinLf0 = 2.995732273553991;

struct Condition {
    size_t sampling_frequency = 0, fperiod = 0;
    double volume = 1.0;
    std::vector<double> msd_threshold, gv_weight;
    bool phoneme_alignment = false;
    bool batch_invariant = false; // jb_engine_set_batch_invariant: JB_BATCH_SERIAL | JB_BATCH_SERIAL_GV for every batch of this engine
    double speed = 1.0;
    size_t stage = 0;
    bool use_log_gain = false;
    double alpha = 0.0, beta = 0.0, additional_half_tone = 0.0;
    // InterporationWeight
    std::vector<double> w_duration;
    std::vector<std::vector<double>> w_param, w_gv;
};

struct Engine {
    std::vector<std::shared_ptr<Voice>> voices;
    Condition cond;
    // cached static description for the state-level ABI
    jb_voice_desc desc{};
    std::vector<double> win_coef[kMaxStream];
    void refresh_desc();
    // stream pdf tables with all trees concatenated (row = tree_off[tree] + pdf-1),
    // built once, and their device copies per GPU
    struct CatTable {
        std::vector<float> rows;
        std::vector<uint32_t> tree_off;
        uint32_t n_rows = 0, row_len = 0;
    };
    mutable std::vector<CatTable> cat; // [voice * nstream + stream]
    mutable std::map<int, jb_pdf_set *> pdf_sets;
    mutable std::mutex pdf_mu;
    int pdf_set_for(int device, const jb_pdf_set **out) const;
    ~Engine()
    {
        for (auto &kv : pdf_sets)
            jb_pdf_set_free(kv.second);
    }
};

int Engine::pdf_set_for(int device, const jb_pdf_set **out) const
{
    std::lock_guard<std::mutex> lk(pdf_mu);
    const size_t ns = std::min(voices[0]->streams.size(), (size_t)kMaxStream);
    if (cat.empty()) {
        cat.resize(voices.size() * ns);
        for (size_t v = 0; v < voices.size(); v++)
            for (size_t si = 0; si < ns; si++) {
                const Model &m = voices[v]->streams[si].stream;
                CatTable &c = cat[v * ns + si];
                c.row_len = (uint32_t)m.pdf_len;
                for (size_t t = 0; t < m.pdf.size(); t++) {
                    c.tree_off.push_back(c.n_rows);
                    c.rows.insert(c.rows.end(), m.pdf[t].begin(), m.pdf[t].end());
                    c.n_rows += (uint32_t)m.npdf[t];
                }
            }
    }
    auto it = pdf_sets.find(device);
    if (it == pdf_sets.end()) {
        std::vector<jb_pdf_table> tabs(cat.size());
        for (size_t i = 0; i < cat.size(); i++)
            tabs[i] = jb_pdf_table{cat[i].rows.data(), cat[i].n_rows, cat[i].row_len};
        jb_pdf_set *ps = nullptr;
        int rc = jb_pdf_set_create(tabs.data(), (uint32_t)voices.size(), (uint32_t)ns, device, &ps);
        if (rc)
            return rc;
        it = pdf_sets.emplace(device, ps).first;
    }
    *out = it->second;
    return JB_OK;
}

void Engine::refresh_desc()
{
    const Voice &v = *voices[0];
    memset(&desc, 0, sizeof desc);
    desc.sampling_frequency = (uint32_t)cond.sampling_frequency;
    desc.fperiod = (uint32_t)cond.fperiod;
    desc.nstream = (uint32_t)v.meta.num_streams;
    desc.stage = (uint32_t)cond.stage;
    desc.use_log_gain = cond.use_log_gain;
    desc.alpha = cond.alpha;
    desc.beta = cond.beta;
    desc.volume = cond.volume;
    for (size_t i = 0; i < v.streams.size() && i < (size_t)kMaxStream; i++) {
        const StreamModel &s = v.streams[i];
        jb_stream_desc &d = desc.stream[i];
        d.vector_length = (uint32_t)s.vector_length;
        d.num_windows = (uint32_t)s.num_windows;
        d.is_msd = s.is_msd;
        d.use_gv = s.use_gv;
        win_coef[i].clear();
        for (size_t w = 0; w < s.windows.size() && w < JB_MAX_WINDOW; w++) {
            d.win_width[w] = (uint32_t)s.windows[w].size();
            win_coef[i].insert(win_coef[i].end(), s.windows[w].begin(), s.windows[w].end());
        }
        d.win_coef = win_coef[i].data();
    }
}

// Condition::load_model (src/engine.rs:84-125)
static int load_condition(Engine &e)
{
    const Voice &v = *e.voices[0];
    Condition &c = e.cond;
    const size_t ns = (size_t)v.meta.num_streams, nv = e.voices.size();
    c.sampling_frequency = (size_t)v.meta.sampling_frequency;
    c.fperiod = (size_t)v.meta.frame_period;
    c.msd_threshold.assign(ns, 0.5);
    c.gv_weight.assign(ns, 1.0);
    for (const std::string &opt : v.streams[0].options) { // options of stream 0 only
        size_t eq = opt.find('=');
        if (eq == std::string::npos)
            continue; // "Skipped unrecognized option"
        std::string key = opt.substr(0, eq), val = opt.substr(eq + 1);
        char *end = nullptr;
        if (key == "GAMMA") {
            unsigned long g = strtoul(val.c_str(), &end, 10);
            if (val.empty() || *end) {
                set_error("Failed to parse option GAMMA");
                return JB_ERR_PARSE_OPTION;
            }
            c.stage = g;
        } else if (key == "LN_GAIN") {
            if (val == "1")
                c.use_log_gain = true;
            else if (val == "0")
                c.use_log_gain = false;
            else {
                set_error("Failed to parse option LN_GAIN");
                return JB_ERR_PARSE_OPTION;
            }
        } else if (key == "ALPHA") {
            double a = strtod(val.c_str(), &end);
            if (val.empty() || *end) {
                set_error("Failed to parse option ALPHA");
                return JB_ERR_PARSE_OPTION;
            }
            c.alpha = a;
        }
    }
    // InterporationWeight::new(nvoices, nstream): equal weights
    c.w_duration.assign(nv, 1.0 / (double)nv);
    c.w_param.assign(ns, c.w_duration);
    c.w_gv.assign(ns, c.w_duration);
    return JB_OK;
}

// VoiceSet::new metadata checks (src/model/voice_set.rs:22-42)
static int check_voiceset(const std::vector<std::shared_ptr<Voice>> &vs)
{
    if (vs.empty()) {
        set_error("No HTS voice was given.");
        return JB_ERR_MODEL;
    }
    const Voice &f = *vs[0];
    for (size_t i = 1; i < vs.size(); i++) {
        const Voice &v = *vs[i];
        bool ok = v.meta == f.meta && v.streams.size() == f.streams.size();
        for (size_t s = 0; ok && s < f.streams.size(); s++) {
            const StreamModel &a = v.streams[s], &b = f.streams[s];
            ok = a.vector_length == b.vector_length && a.num_windows == b.num_windows &&
                 a.is_msd == b.is_msd && a.use_gv == b.use_gv && a.options == b.options;
        }
        if (!ok) {
            set_error("The global metadata does not match.");
            return JB_ERR_MODEL;
        }
    }
    return JB_OK;
}

// ---- labels ----------------------------------------------------------------
struct ParsedLabels {
    std::vector<std::string> labels;
    std::vector<std::pair<double, double>> times;
};

// minimal shape check standing in for jlabel's grammar (src/label.rs:65,72)
static bool label_shape_ok(const std::string &l)
{
    return glob_match("*^*-*+*=*/A:*/B:*/C:*/D:*/E:*/F:*/G:*/H:*/I:*/J:*/K:*", l);
}

static int parse_labels(const Condition &c, const char *const *lines, size_t n, ParsedLabels &out)
{
    const double rate = (double)c.sampling_frequency / ((double)c.fperiod * 1e+7);
    for (size_t i = 0; i < n; i++) {
        if (!lines[i]) {
            set_error("null label line");
            return JB_ERR_LABEL;
        }
        std::string line(lines[i]);
        size_t s1 = line.find(' ');
        if (s1 != std::string::npos) {
            size_t s2 = line.find(' ', s1 + 1);
            if (s2 == std::string::npos) {
                set_error("Expected a fullcontext-label in " + line);
                return JB_ERR_LABEL;
            }
            char *e1 = nullptr, *e2 = nullptr;
            std::string a = line.substr(0, s1), b = line.substr(s1 + 1, s2 - s1 - 1);
            double st = strtod(a.c_str(), &e1), en = strtod(b.c_str(), &e2);
            if (a.empty() || b.empty() || *e1 || *e2) {
                set_error("Failed to parse as floating-point number");
                return JB_ERR_LABEL;
            }
            out.times.emplace_back(st * rate, en * rate);
            out.labels.push_back(line.substr(s2 + 1));
        } else if (line.empty()) {
            continue;
        } else {
            out.times.emplace_back(-1.0, -1.0);
            out.labels.push_back(line);
        }
        if (!label_shape_ok(out.labels.back())) {
            set_error("jlabel failed to parse fullcontext-label: " + out.labels.back());
            return JB_ERR_LABEL;
        }
    }
    // Labels::new (src/label.rs:83-113)
    auto &t = out.times;
    for (size_t i = 0; i < t.size(); i++) {
        if (i + 1 < t.size()) {
            if (t[i].second < 0.0 && t[i + 1].first >= 0.0)
                t[i].second = t[i + 1].first;
            else if (t[i].second >= 0.0 && t[i + 1].first < 0.0)
                t[i + 1].first = t[i].second;
        }
        if (t[i].first < 0.0)
            t[i].first = -1.0;
        if (t[i].second < 0.0)
            t[i].second = -1.0;
    }
    return JB_OK;
}

// ---- durations (src/duration.rs) -------------------------------------------
struct MV {
    double mean, vari;
};

static void estimate_duration(const MV *p, size_t n, double rho, uint32_t *d)
{
    for (size_t i = 0; i < n; i++) {
        double r = std::round(p[i].mean + rho * p[i].vari); // f64::round: half away from zero
        d[i] = (uint32_t)(r > 1.0 ? r : 1.0);
    }
}

static void estimate_with_frame_length(const MV *p, size_t n, double frame_length, uint32_t *d)
{
    double tl = std::round(frame_length);
    const size_t target = (size_t)(tl > 1.0 ? tl : 1.0);
    if (target <= n) {
        std::fill(d, d + n, 1u);
        return;
    }
    double mean = 0.0, vari = 0.0;
    for (size_t i = 0; i < n; i++) {
        mean = mean + p[i].mean;
        vari = vari + p[i].vari;
    }
    const double rho = ((double)target - mean) / vari;
    estimate_duration(p, n, rho, d);
    if (n == 0)
        return;
    size_t sum = 0;
    for (size_t i = 0; i < n; i++)
        sum += d[i];
    auto cost = [&](double dd, const MV &q) { return std::fabs(rho - (dd - q.mean) / q.vari); };
    while (sum != target) {
        const bool grow = target > sum;
        size_t best = n;
        double bc = 0.0;
        for (size_t i = 0; i < n; i++) {
            if (!grow && d[i] <= 1)
                continue;
            double c = cost((double)d[i] + (grow ? 1.0 : -1.0), p[i]);
            if (best == n || c < bc) { // first minimum wins (Iterator::min_by)
                best = i;
                bc = c;
            }
        }
        if (best == n)
            break;
        if (grow) {
            d[best]++;
            sum++;
        } else {
            d[best]--;
            sum--;
        }
    }
}

// ---- state construction ------------------------------------------------------
struct States {
    jb_state_utt utt{};
    // indexed form: pdf rows per stream and voice instead of blended Gaussians
    jb_index_utt iutt{};
    std::vector<uint32_t> rows[kMaxStream][JB_MAX_VOICES];
    std::vector<double> weights[kMaxStream];
    std::vector<uint32_t> dur;
    std::vector<double> dur_pdf; // [S][2]: Models::duration (model/mod.rs:80-92), the blended (mean, variance) the durations come from
    std::vector<double> mean[kMaxStream], var[kMaxStream], msd[kMaxStream], gvm[kMaxStream],
        gvv[kMaxStream];
    std::vector<uint8_t> gsw[kMaxStream];
};

// VoiceSet::weighted (voice_set.rs:80-95): first*w0, then += w_i * param_i, in order.
template <class F>
static void blend(const Engine &e, const std::vector<double> &w, size_t len, double *out, F get)
{
    const float *p0 = get(*e.voices[0]);
    for (size_t k = 0; k < len; k++)
        out[k] = (double)p0[k] * w[0];
    for (size_t v = 1; v < e.voices.size(); v++) {
        const float *p = get(*e.voices[v]);
        for (size_t k = 0; k < len; k++)
            out[k] += w[v] * (double)p[k];
    }
}

// indexed == true: the stream Gaussians are NOT blended on the host; st.iutt carries the pdf rows
// (tree search result) of every state and voice for the device-side gather (needs Engine::cat).
// Runs fn(lo, hi) over [0, n) in `nt` contiguous pieces on host threads (in the calling thread when nt == 1);
// a ModelError thrown by a piece is rethrown here.  The per-label work of the front half (tree searches with
// string-predicate questions, pdf blend) is independent per label and read-only on the engine.
template <class F> static void parallel_labels(size_t n, unsigned nt, F fn)
{
    nt = (unsigned)std::min<size_t>(nt ? nt : 1u, std::max<size_t>(n / 32, 1)); // at least 32 labels per thread
    if (nt <= 1) {
        fn((size_t)0, n);
        return;
    }
    std::vector<std::string> errs(nt);
    std::vector<uint8_t> failed(nt, 0);
    std::vector<std::thread> pool;
    for (unsigned t = 0; t < nt; t++)
        pool.emplace_back([&, t] {
            try {
                fn(n * t / nt, n * (t + 1) / nt);
            } catch (const ModelError &ex) {
                failed[t] = 1;
                errs[t] = ex.what();
            }
        });
    for (auto &th : pool)
        th.join();
    for (unsigned t = 0; t < nt; t++)
        if (failed[t])
            throw ModelError(errs[t]);
}

// label_threads: host threads for the per-label work of ONE utterance (1 inside jb_synthesize_batch, whose
// workers already take an utterance each; several for the single-utterance entries, where a 1,456-label text
// otherwise spends 15-20 ms in tree searches before the device sees anything)
// worker threads of the front half: JB_HOST_THREADS, default min(16, cores)
static unsigned host_threads_default()
{
    unsigned nt = std::thread::hardware_concurrency();
    nt = nt ? std::min(nt, 16u) : 1u;
    if (const char *ev = getenv("JB_HOST_THREADS"))
        nt = (unsigned)std::max(1, atoi(ev));
    return nt;
}

static int build_states(const Engine &e, const char *const *lines, size_t n, States &st, bool indexed = false,
                        unsigned label_threads = 1)
{
    const Condition &c = e.cond;
    const Voice &v0 = *e.voices[0];
    // JB_FRONT_TRACE=1: phases of the front half of one utterance on stderr (like JB_CREATE_TRACE / JB_REDO_TRACE)
    static const bool trace = getenv("JB_FRONT_TRACE") && atoi(getenv("JB_FRONT_TRACE")) != 0;
    auto t_prev = std::chrono::steady_clock::now();
    auto fmark = [&](const char *what) {
        if (!trace)
            return;
        const auto t = std::chrono::steady_clock::now();
        fprintf(stderr, "front half: %-28s %8.3f ms (%zu labels, %u threads)\n", what,
                std::chrono::duration<double, std::milli>(t - t_prev).count(), n, label_threads);
        t_prev = t;
    };
    ParsedLabels pl;
    int rc = parse_labels(c, lines, n, pl);
    if (rc)
        return rc;
    fmark("labels parsed");
    const size_t nl = pl.labels.size(), ns = (size_t)v0.meta.num_states, S = nl * ns;
    st.dur.assign(S, 0);
    try {
        // Models::duration (model/mod.rs:80-92) and Models::stream (:98-118) of every label in ONE parallel pass
        // (round 5: four passes, each starting and joining its own threads, were most of the front half of a
        // 200-label text -- 1.2 ms, of which the tree searches are 0.2 on six threads)
        std::vector<MV> dp(S);
        const size_t nsx = std::min(v0.streams.size(), (size_t)kMaxStream);
        size_t plen_max = 0;
        for (size_t si = 0; si < nsx; si++) {
            const StreamModel &sm = v0.streams[si];
            const size_t WL = (size_t)sm.vector_length * (size_t)sm.num_windows;
            plen_max = std::max(plen_max, 2 * WL + (sm.is_msd ? 1 : 0));
            if (indexed) {
                for (size_t v = 0; v < e.voices.size(); v++)
                    st.rows[si][v].assign(S, 0);
                st.weights[si] = c.w_param[si];
            } else {
                st.mean[si].assign(S * WL, 0.0);
                st.var[si].assign(S * WL, 0.0);
                st.msd[si].assign(S, DBL_MAX);
            }
        }
        parallel_labels(nl, label_threads, [&](size_t lo, size_t hi) {
            std::vector<double> tmp(2 * ns), buf(plen_max);
            QuestionMemo memo; // question results of the current label, per model
            for (size_t i = lo; i < hi; i++) {
                memo.reset();
                blend(e, c.w_duration, 2 * ns, tmp.data(),
                      [&](const Voice &v) { return v.duration.get_parameter(2, pl.labels[i], &memo); });
                for (size_t s = 0; s < ns; s++)
                    dp[i * ns + s] = {tmp[s], tmp[s + ns]};
                for (size_t si = 0; si < nsx; si++) {
                    const StreamModel &sm = v0.streams[si];
                    const size_t WL = (size_t)sm.vector_length * (size_t)sm.num_windows;
                    const size_t plen = 2 * WL + (sm.is_msd ? 1 : 0);
                    for (size_t s = 0; s < ns; s++) {
                        const size_t row = i * ns + s;
                        if (indexed) {
                            for (size_t v = 0; v < e.voices.size(); v++) {
                                const Model &m = e.voices[v]->streams[si].stream;
                                int tp, pi;
                                m.get_index((int)(2 + s), pl.labels[i], tp, pi, &memo);
                                if (tp < 0 || pi < 1 || pi > m.npdf[(size_t)tp])
                                    throw ModelError("index not found"); // reference: todo!() (voice/model.rs:76-79)
                                st.rows[si][v][row] = e.cat[v * nsx + si].tree_off[(size_t)tp] + (uint32_t)(pi - 1);
                            }
                        } else {
                            blend(e, c.w_param[si], plen, buf.data(), [&](const Voice &v) {
                                return v.streams[si].stream.get_parameter((int)(2 + s), pl.labels[i], &memo);
                            });
                            std::copy(buf.begin(), buf.begin() + WL, st.mean[si].begin() + row * WL);
                            std::copy(buf.begin() + WL, buf.begin() + 2 * WL, st.var[si].begin() + row * WL);
                            if (sm.is_msd)
                                st.msd[si][row] = buf[2 * WL];
                        }
                    }
                }
            }
        });
        fmark("pdfs of all models");
        st.dur_pdf.resize(2 * S);
        for (size_t s = 0; s < S; s++) {
            st.dur_pdf[2 * s] = dp[s].mean;
            st.dur_pdf[2 * s + 1] = dp[s].vari;
        }
        if (S) {
            if (c.phoneme_alignment) {
                // create_with_alignment (duration.rs:41-65)
                size_t frame_count = 0, next_state = 0, state = 0, nd = 0;
                for (size_t i = 0; i < nl; i++) {
                    double end_frame = pl.times[i].second;
                    if (end_frame >= 0.0) {
                        size_t cnt = state + ns - next_state;
                        estimate_with_frame_length(dp.data() + next_state, cnt,
                                                   end_frame - (double)frame_count, st.dur.data() + nd);
                        for (size_t k = 0; k < cnt; k++)
                            frame_count += st.dur[nd + k];
                        nd += cnt;
                        next_state = state + ns;
                    }
                    state += ns;
                }
                // states after the last aligned label get no duration in the reference
            } else {
                // create (duration.rs:28-38)
                estimate_duration(dp.data(), S, 0.0, st.dur.data());
                if (c.speed != 1.0) {
                    size_t length = 0;
                    for (uint32_t d : st.dur)
                        length += d;
                    estimate_with_frame_length(dp.data(), S, (double)length / c.speed, st.dur.data());
                }
            }
        }
        fmark("durations");
        st.utt.num_states = (uint32_t)S;
        st.utt.durations = st.dur.data();
        // Models::stream / gv (model/mod.rs:98-146)
        for (size_t si = 0; si < v0.streams.size() && si < (size_t)kMaxStream; si++) {
            const StreamModel &sm = v0.streams[si];
            if (indexed) {
                jb_index_stream &io = st.iutt.stream[si];
                for (size_t v = 0; v < e.voices.size(); v++)
                    io.row[v] = st.rows[si][v].data();
                io.weight = st.weights[si].data();
            }
            jb_stream_states &o = st.utt.stream[si];
            o.mean = st.mean[si].data();
            o.var = st.var[si].data();
            o.msd = sm.is_msd ? st.msd[si].data() : nullptr;
            o.gv_weight = c.gv_weight[si];
            o.msd_threshold = c.msd_threshold[si];
            if (sm.use_gv && nl > 0) {
                const size_t L = (size_t)sm.vector_length;
                std::vector<double> g(2 * L);
                blend(e, c.w_gv[si], 2 * L, g.data(), [&](const Voice &v) {
                    return v.streams[si].gv->get_parameter(2, pl.labels[0]); // first label only
                });
                st.gvm[si].assign(g.begin(), g.begin() + L);
                st.gvv[si].assign(g.begin() + L, g.end());
                st.gsw[si].assign(S, 0);
                for (size_t i = 0; i < nl; i++) {
                    uint8_t sw = !v0.gv_off.test(pl.labels[i]);
                    for (size_t s = 0; s < ns; s++)
                        st.gsw[si][i * ns + s] = sw;
                }
                o.gv_mean = st.gvm[si].data();
                o.gv_var = st.gvv[si].data();
                o.gv_switch = st.gsw[si].data();
            }
                    if (indexed) {
                jb_index_stream &io = st.iutt.stream[si];
                io.gv_mean = o.gv_mean;
                io.gv_var = o.gv_var;
                io.gv_switch = o.gv_switch;
                io.gv_weight = o.gv_weight;
                io.msd_threshold = o.msd_threshold;
            }
        }
        if (indexed) {
            st.iutt.num_states = (uint32_t)S;
            st.iutt.durations = st.dur.data();
            st.iutt.lf0_offset = c.additional_half_tone * kHalfTone;
        }
        // apply_additional_half_tone (stream_parameter.rs:29-37, engine.rs:342-345)
        if (!indexed && c.additional_half_tone != 0.0 && v0.streams.size() > 1) {
            const size_t WL = (size_t)v0.streams[1].vector_length * (size_t)v0.streams[1].num_windows;
            for (size_t s = 0; s < S; s++) {
                double x = st.mean[1][s * WL] + c.additional_half_tone * kHalfTone;
                st.mean[1][s * WL] = std::min(std::max(x, kMinLf0), kMaxLf0);
            }
        }
    } catch (const ModelError &ex) {
        set_error(std::string("Model error: ") + ex.what());
        return JB_ERR_MODEL;
    }
    return JB_OK;
}

// SpeechGenerator (src/speech.rs:9-96).  The reference's generator owns its three parameter tracks and a
// Vocoder and advances one frame per generate_step; nothing can change between steps, so what a step
// returns is fixed when the generator is made.  Here the whole utterance is put on the device's queue at
// creation (parameter generation, the time-chunked vocoder, the hand-off certification: the path of
// jb_synthesize) and NOT waited for; a step hands out the next frame(s) of the finished PCM from a block
// cache on the host.  Until that run has finished, the first kGenSerialFrames steps are served by the
// serial recursion one frame at a time on a side stream (persistent filter / excitation state in
// vd.state), so that the first samples arrive without waiting for the last ones.
struct Generator {
    std::unique_ptr<Batch> batch;
    size_t fperiod = 0, next = 0, total = 0;
    bool ahead_ready = false;   // the whole-utterance run is finished and certified
    bool serial_armed = false;  // the side stream waits for parameter generation
    std::vector<double> cache;  // PCM of frames [cache_first, cache_first + cache_frames)
    size_t cache_first = 0, cache_frames = 0;
};
constexpr size_t kGenSerialFrames = 8;  // steps that may be served serially while the utterance is in flight
constexpr size_t kGenBlockFrames = 256; // frames per D2H block of the cache (480 KB at 240 samples per frame)

} // namespace jb

using namespace jb;

#define ENG(e) ((jb::Engine *)(e))
#define CENG(e) ((const jb::Engine *)(e))

extern "C" {

static int finish_load(std::unique_ptr<jb::Engine> &e, jb_engine **out)
{
    int rc = check_voiceset(e->voices);
    if (rc)
        return rc;
    if ((rc = load_condition(*e)))
        return rc;
    e->refresh_desc();
    *out = (jb_engine *)e.release();
    return JB_OK;
}

int jb_engine_load(const char *const *paths, size_t n, jb_engine **out)
{
    if (!out || (n && !paths))
        return JB_ERR_INVALID;
    *out = nullptr;
    std::unique_ptr<jb::Engine> e(new jb::Engine());
    try {
        for (size_t i = 0; i < n; i++)
            e->voices.push_back(load_htsvoice(paths[i]));
    } catch (const std::exception &ex) {
        set_error(std::string("Model error: ") + ex.what());
        return JB_ERR_MODEL;
    }
    return finish_load(e, out);
}

int jb_engine_load_from_bytes(const uint8_t *const *bufs, const size_t *lens, size_t n, jb_engine **out)
{
    if (!out || (n && (!bufs || !lens)))
        return JB_ERR_INVALID;
    *out = nullptr;
    std::unique_ptr<jb::Engine> e(new jb::Engine());
    try {
        for (size_t i = 0; i < n; i++)
            e->voices.push_back(parse_htsvoice(bufs[i], lens[i]));
    } catch (const std::exception &ex) {
        set_error(std::string("Model error: ") + ex.what());
        return JB_ERR_MODEL;
    }
    return finish_load(e, out);
}

// Engine::new(voices, condition) (src/engine.rs:289-291).  The reference's VoiceSet holds Arc<Voice>
// (voice_set.rs:17): the new engine SHARES the voices of `voices_of` and takes a COPY of the Condition of
// `condition_of`.  With both arguments the same engine this is Engine::clone (engine.rs:246).
int jb_engine_new(const jb_engine *voices_of, const jb_engine *condition_of, jb_engine **out)
{
    if (!out || !voices_of || !condition_of)
        return JB_ERR_INVALID;
    *out = nullptr;
    const jb::Engine *a = CENG(voices_of), *c = CENG(condition_of);
    // a Condition made for another voice set: the interpolation weights and the per-stream arrays must fit
    if (c->cond.w_duration.size() != a->voices.size() || c->cond.w_param.size() != a->voices[0]->streams.size() ||
        c->cond.msd_threshold.size() != a->voices[0]->streams.size()) {
        set_error("Weights length is invalid; the condition was made for another voice set");
        return JB_ERR_WEIGHT;
    }
    std::unique_ptr<jb::Engine> e(new jb::Engine());
    e->voices = a->voices;
    e->cond = c->cond;
    e->refresh_desc();
    *out = (jb_engine *)e.release();
    return JB_OK;
}

void jb_engine_free(jb_engine *e) { delete ENG(e); }

// ---- Condition (src/engine.rs:127-243) ----
int jb_engine_set_sampling_frequency(jb_engine *e, size_t v)
{
    ENG(e)->cond.sampling_frequency = std::max<size_t>(v, 1);
    ENG(e)->refresh_desc();
    return JB_OK;
}
size_t jb_engine_get_sampling_frequency(const jb_engine *e) { return CENG(e)->cond.sampling_frequency; }
int jb_engine_set_fperiod(jb_engine *e, size_t v)
{
    ENG(e)->cond.fperiod = std::max<size_t>(v, 1);
    ENG(e)->refresh_desc();
    return JB_OK;
}
size_t jb_engine_get_fperiod(const jb_engine *e) { return CENG(e)->cond.fperiod; }
int jb_engine_set_volume(jb_engine *e, double db)
{
    ENG(e)->cond.volume = std::exp(db * kDB);
    ENG(e)->refresh_desc();
    return JB_OK;
}
double jb_engine_get_volume(const jb_engine *e) { return std::log(CENG(e)->cond.volume) / kDB; }
int jb_engine_set_msd_threshold(jb_engine *e, size_t s, double v)
{
    if (s >= ENG(e)->cond.msd_threshold.size())
        return JB_ERR_INVALID;
    ENG(e)->cond.msd_threshold[s] = std::min(std::max(v, 0.0), 1.0);
    return JB_OK;
}
double jb_engine_get_msd_threshold(const jb_engine *e, size_t s)
{
    return s < CENG(e)->cond.msd_threshold.size() ? CENG(e)->cond.msd_threshold[s] : NAN;
}
int jb_engine_set_gv_weight(jb_engine *e, size_t s, double v)
{
    if (s >= ENG(e)->cond.gv_weight.size())
        return JB_ERR_INVALID;
    ENG(e)->cond.gv_weight[s] = std::max(v, 0.0);
    return JB_OK;
}
double jb_engine_get_gv_weight(const jb_engine *e, size_t s)
{
    return s < CENG(e)->cond.gv_weight.size() ? CENG(e)->cond.gv_weight[s] : NAN;
}
int jb_engine_set_phoneme_alignment_flag(jb_engine *e, int f)
{
    ENG(e)->cond.phoneme_alignment = f != 0;
    return JB_OK;
}
int jb_engine_get_phoneme_alignment_flag(const jb_engine *e) { return CENG(e)->cond.phoneme_alignment; }
int jb_engine_set_batch_invariant(jb_engine *e, int f)
{
    ENG(e)->cond.batch_invariant = f != 0;
    return JB_OK;
}
int jb_engine_get_batch_invariant(const jb_engine *e) { return CENG(e)->cond.batch_invariant; }
int jb_engine_set_speed(jb_engine *e, double v)
{
    ENG(e)->cond.speed = std::max(v, 1.0E-06);
    return JB_OK;
}
double jb_engine_get_speed(const jb_engine *e) { return CENG(e)->cond.speed; }
int jb_engine_set_alpha(jb_engine *e, double v)
{
    ENG(e)->cond.alpha = std::min(std::max(v, 0.0), 1.0);
    ENG(e)->refresh_desc();
    return JB_OK;
}
double jb_engine_get_alpha(const jb_engine *e) { return CENG(e)->cond.alpha; }
int jb_engine_set_beta(jb_engine *e, double v)
{
    ENG(e)->cond.beta = std::min(std::max(v, 0.0), 1.0);
    ENG(e)->refresh_desc();
    return JB_OK;
}
double jb_engine_get_beta(const jb_engine *e) { return CENG(e)->cond.beta; }
int jb_engine_set_additional_half_tone(jb_engine *e, double v)
{
    ENG(e)->cond.additional_half_tone = v;
    return JB_OK;
}
double jb_engine_get_additional_half_tone(const jb_engine *e) { return CENG(e)->cond.additional_half_tone; }
size_t jb_engine_num_voices(const jb_engine *e) { return CENG(e)->voices.size(); }
size_t jb_engine_num_streams(const jb_engine *e) { return (size_t)CENG(e)->voices[0]->meta.num_streams; }
size_t jb_engine_num_states(const jb_engine *e) { return (size_t)CENG(e)->voices[0]->meta.num_states; }

int jb_engine_set_interpolation_weight(jb_engine *e, int which, size_t stream, const double *w, size_t n)
{
    Condition &c = ENG(e)->cond;
    if (!w)
        return JB_ERR_INVALID;
    // Weights::new: sum must equal 1.0 within f64::EPSILON (approx default)
    double sum = 0.0;
    for (size_t i = 0; i < n; i++)
        sum += w[i];
    if (std::fabs(sum - 1.0) > DBL_EPSILON) {
        set_error("Weights do not sum to 1.0");
        return JB_ERR_WEIGHT;
    }
    if (n != ENG(e)->voices.size()) {
        set_error("Weights length is invalid; expected " + std::to_string(ENG(e)->voices.size()) +
                  ", got " + std::to_string(n));
        return JB_ERR_WEIGHT;
    }
    std::vector<double> v(w, w + n);
    if (which == 0) {
        c.w_duration = v;
    } else if (which == 1 || which == 2) {
        auto &tab = which == 1 ? c.w_param : c.w_gv;
        if (stream >= tab.size())
            return JB_ERR_INVALID; // reference: index panic
        tab[stream] = v;
    } else {
        return JB_ERR_INVALID;
    }
    return JB_OK;
}

// InterporationWeight::{get_duration, get_parameter, get_gv} (src/model/interporation_weight.rs:115-125)
int jb_engine_get_interpolation_weight(const jb_engine *e, int which, size_t stream, double *w, size_t cap, size_t *n)
{
    const Condition &c = CENG(e)->cond;
    const std::vector<double> *v = nullptr;
    if (which == 0)
        v = &c.w_duration;
    else if (which == 1 || which == 2) {
        const auto &tab = which == 1 ? c.w_param : c.w_gv;
        if (stream >= tab.size())
            return JB_ERR_INVALID; // reference: index panic
        v = &tab[stream];
    } else {
        return JB_ERR_INVALID;
    }
    if (n)
        *n = v->size();
    if (!w)
        return JB_OK;
    if (cap < v->size())
        return JB_ERR_BUFFER;
    std::copy(v->begin(), v->end(), w);
    return JB_OK;
}

// ---- model introspection ----
static const jb::Model *model_of(const jb::Engine *e, size_t voice, int kind)
{
    if (voice >= e->voices.size())
        return nullptr;
    const Voice &v = *e->voices[voice];
    if (kind == 0)
        return &v.duration;
    if (kind >= 1 && kind <= 3 && (size_t)(kind - 1) < v.streams.size())
        return &v.streams[(size_t)(kind - 1)].stream;
    if (kind >= 4 && kind <= 6 && (size_t)(kind - 4) < v.streams.size() &&
        v.streams[(size_t)(kind - 4)].gv)
        return &*v.streams[(size_t)(kind - 4)].gv;
    return nullptr;
}

int jb_engine_model_shape(const jb_engine *e, size_t voice, int kind, size_t *ntree, size_t *pdf_len)
{
    const jb::Model *m = model_of(CENG(e), voice, kind);
    if (!m)
        return JB_ERR_INVALID;
    if (ntree)
        *ntree = m->trees.size();
    if (pdf_len)
        *pdf_len = (size_t)m->pdf_len;
    return JB_OK;
}

int jb_engine_pdf_table(const jb_engine *e, size_t voice, int kind, size_t tree, const float **table,
                        size_t *npdf)
{
    const jb::Model *m = model_of(CENG(e), voice, kind);
    if (!m || tree >= m->trees.size() || !table)
        return JB_ERR_INVALID;
    *table = m->pdf[tree].data();
    if (npdf)
        *npdf = (size_t)m->npdf[tree];
    return JB_OK;
}

int jb_engine_tree_index(const jb_engine *e, size_t voice, int kind, int state_index, const char *label,
                         int *tree_state, int *pdf_index)
{
    const jb::Model *m = model_of(CENG(e), voice, kind);
    if (!m || !label)
        return JB_ERR_INVALID;
    int tp, pi;
    m->get_index(state_index, label, tp, pi);
    if (tree_state)
        *tree_state = tp < 0 ? -1 : m->trees[(size_t)tp].state;
    if (pdf_index)
        *pdf_index = pi;
    return JB_OK;
}

// ---- states ----
int jb_engine_states(const jb_engine *e, const char *const *lines, size_t n, jb_states **out)
{
    if (!e || !out || (n && !lines))
        return JB_ERR_INVALID;
    *out = nullptr;
    std::unique_ptr<jb::States> st(new jb::States());
    int rc = build_states(*CENG(e), lines, n, *st, false, host_threads_default());
    if (rc)
        return rc;
    *out = (jb_states *)st.release();
    return JB_OK;
}
const jb_state_utt *jb_states_utt(const jb_states *s) { return s ? &((const jb::States *)s)->utt : nullptr; }
const double *jb_states_duration_params(const jb_states *s)
{
    return s && !((const jb::States *)s)->dur_pdf.empty() ? ((const jb::States *)s)->dur_pdf.data() : nullptr;
}
const jb_voice_desc *jb_engine_voice_desc(const jb_engine *e) { return e ? &CENG(e)->desc : nullptr; }
void jb_states_free(jb_states *s) { delete (jb::States *)s; }

// ---- synthesize ----
void jb_pcm_free(double *p) { free(p); }

// ---- WAV sink (examples/is-bonsai/main.rs:37-49) ----
static int write_wav(const char *path, const int16_t *pcm, size_t n, uint32_t fs)
{
    if (!path || (!pcm && n) || n > (0xffffffffull - 36) / 2) {
        jb::set_error("bad WAV arguments");
        return JB_ERR_INVALID;
    }
    FILE *f = fopen(path, "wb");
    if (!f) {
        jb::set_error(std::string("cannot open ") + path + ": " + strerror(errno));
        return JB_ERR_MODEL;
    }
    const uint32_t data = (uint32_t)(n * 2), riff = 36 + data, fmt_len = 16, byte_rate = fs * 2;
    const uint16_t pcm_tag = 1, ch = 1, align = 2, bits = 16;
    bool ok = fwrite("RIFF", 1, 4, f) == 4 && fwrite(&riff, 4, 1, f) == 1 && fwrite("WAVEfmt ", 1, 8, f) == 8 &&
              fwrite(&fmt_len, 4, 1, f) == 1 && fwrite(&pcm_tag, 2, 1, f) == 1 && fwrite(&ch, 2, 1, f) == 1 &&
              fwrite(&fs, 4, 1, f) == 1 && fwrite(&byte_rate, 4, 1, f) == 1 && fwrite(&align, 2, 1, f) == 1 &&
              fwrite(&bits, 2, 1, f) == 1 && fwrite("data", 1, 4, f) == 4 && fwrite(&data, 4, 1, f) == 1 &&
              (n == 0 || fwrite(pcm, 2, n, f) == n); // little-endian host (x86-64)
    ok = (fclose(f) == 0) && ok;
    if (!ok) {
        jb::set_error(std::string("short write to ") + path);
        return JB_ERR_MODEL;
    }
    return JB_OK;
}

int jb_write_wav_i16(const char *path, const int16_t *pcm, size_t n, uint32_t fs) { return write_wav(path, pcm, n, fs); }

int jb_write_wav_f64(const char *path, const double *pcm, size_t n, uint32_t fs)
{
    if (!pcm && n)
        return JB_ERR_INVALID;
    std::vector<int16_t> q(n);
    for (size_t i = 0; i < n; i++) {
        double v = std::fmin(pcm[i], 32767.0);
        v = std::fmax(v, -32768.0);
        q[i] = (int16_t)(int)v; // `as i16`: truncation toward zero
    }
    return write_wav(path, q.data(), n, fs);
}

} // extern "C"

// elem = 8: f64 PCM (Engine::synthesize's Vec<f64>); elem = 2: the fused 16-bit sink
int jb::synthesize_batch_impl(const jb_engine *e, const char *const *lines, const size_t *line_off, size_t n_utts,
                              int32_t device, size_t elem, void **pcm, size_t *n_samples, unsigned host_threads)
{
    if (!e || !pcm || !n_samples || (n_utts && !line_off))
        return JB_ERR_INVALID;
    for (size_t u = 0; u < n_utts; u++) {
        pcm[u] = nullptr;
        n_samples[u] = 0;
    }
    std::vector<std::unique_ptr<jb::States>> sts(n_utts);
    for (size_t u = 0; u < n_utts; u++)
        sts[u].reset(new jb::States());
    // JB_E2E_TIMING=1: time spent in the four phases on stderr (tools/e2e_engine.py); with more than
    // one group they overlap, so their sum exceeds the wall time
    const bool timing = getenv("JB_E2E_TIMING") && atoi(getenv("JB_E2E_TIMING")) != 0;
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b2) {
        return std::chrono::duration<double, std::milli>(b2 - a).count();
    };
    const auto t_start = now();
    // The front half (label parse, tree search, pdf blend, durations: label.rs, model/mod.rs:80-156,
    // duration.rs) is independent per utterance and read-only on the engine: host threads, one
    // utterance at a time each (JB_HOST_THREADS, default min(16, cores)).  It is ~13 ms per 128 s
    // utterance and thread against ~0.7 ms of GPU time.
    unsigned nt = host_threads_default();
    if (host_threads && !getenv("JB_HOST_THREADS")) // a multi-device call shares the host cores between its device threads
        nt = host_threads;
    // device-side gather + blend unless JB_HOST_BLEND=1 (A/B: both produce the same bits)
    const bool host_blend = getenv("JB_HOST_BLEND") && atoi(getenv("JB_HOST_BLEND")) != 0;
    const jb_pdf_set *pset = nullptr;
    if (!host_blend) {
        int dev = device;
        if (dev < 0 && hipGetDevice(&dev) != hipSuccess) {
            jb::set_error("no HIP device");
            return JB_ERR_DEVICE;
        }
        int prc = CENG(e)->pdf_set_for(dev, &pset);
        if (prc)
            return prc;
        device = dev;
    }
    const bool indexed = pset != nullptr;

    // A large request goes through in two groups: while the GPU works on the first, the host threads
    // run the front half of the second.  With the 16-bit sink the read-back of the first group also
    // runs beside the GPU work of the second (134 -> 113 ms for 64 x 157 s); with f64 PCM a read-back
    // beside running kernels slows down by more than the overlap gains (203 -> 242 ms), so there all
    // GPU work comes first and the read-backs after it (203-224 -> 183 ms).  More groups lose (three:
    // 220 ms): smaller batches fill the chip less well and their blocks miss the pool.
    // Groups split the utterances by label count; JB_SYNTH_GROUPS overrides.
    size_t ngroups = 1;
    const size_t total_lines = n_utts ? line_off[n_utts] - line_off[0] : 0;
    if (n_utts >= 8 && total_lines >= 16000)
        ngroups = 2;
    if (const char *ev = getenv("JB_SYNTH_GROUPS"))
        ngroups = (size_t)std::max(1, atoi(ev));
    ngroups = std::max<size_t>(1, std::min(ngroups, n_utts));
    std::vector<size_t> glo(ngroups + 1, n_utts);
    glo[0] = 0;
    for (size_t g = 1, u = 0; g < ngroups; g++) {
        const size_t want = line_off[0] + total_lines * g / ngroups;
        while (u < n_utts && line_off[u] < want)
            u++;
        glo[g] = std::max(u, glo[g - 1]);
    }
    std::vector<std::unique_ptr<jb::Batch>> batches(ngroups);
    double t_front = 0, t_create = 0, t_wait = 0, t_d2h = 0;

    auto front = [&](size_t lo, size_t hi) -> int {
        const auto t0 = now();
        std::vector<int> rcs(hi - lo, JB_OK);
        std::vector<std::string> errs(hi - lo);
        std::atomic<size_t> next{lo};
        // fewer utterances than threads (a single long text through jb_synthesize): the spare threads split
        // each utterance's labels
        const unsigned per_utt = (unsigned)std::max<size_t>(1, nt / std::max<size_t>(1, hi - lo));
        auto work = [&]() {
            for (size_t u; (u = next.fetch_add(1)) < hi;) {
                rcs[u - lo] = build_states(*CENG(e), lines + line_off[u], line_off[u + 1] - line_off[u], *sts[u], indexed,
                                           per_utt);
                if (rcs[u - lo])
                    errs[u - lo] = jb::g_err; // the worker's thread-local message
            }
        };
        const unsigned n = (unsigned)std::min<size_t>(nt, hi - lo);
        if (n <= 1) {
            work();
        } else {
            std::vector<std::thread> pool;
            for (unsigned k = 0; k < n; k++)
                pool.emplace_back(work);
            for (auto &t : pool)
                t.join();
        }
        t_front += ms(t0, now());
        for (size_t u = lo; u < hi; u++)
            if (rcs[u - lo]) {
                jb::set_error(errs[u - lo]);
                return rcs[u - lo];
            }
        return JB_OK;
    };
    auto launch = [&](size_t g) -> int {
        const auto t0 = now();
        const size_t lo = glo[g], hi = glo[g + 1];
        jb_batch_opts opts{};
        opts.device = device;
        opts.flags = (elem == 2 ? JB_BATCH_PCM_I16 : 0) | (CENG(e)->cond.batch_invariant ? (JB_BATCH_SERIAL | JB_BATCH_SERIAL_GV) : 0);
        jb::Batch *b = nullptr;
        int rc;
        if (indexed) {
            std::vector<jb_index_utt> iu(hi - lo);
            for (size_t u = lo; u < hi; u++)
                iu[u - lo] = sts[u]->iutt;
            rc = jb_batch_create_indexed(&CENG(e)->desc, pset, iu.data(), hi - lo, &opts, (jb_batch **)&b);
        } else {
            std::vector<jb_state_utt> su(hi - lo);
            for (size_t u = lo; u < hi; u++)
                su[u - lo] = sts[u]->utt;
            rc = jb::Batch::create(&CENG(e)->desc, su.data(), hi - lo, &opts, &b);
        }
        if (rc)
            return rc;
        batches[g].reset(b);
        rc = b->run(false);
        t_create += ms(t0, now());
        return rc;
    };
    auto finish_sync = [&](size_t g) -> int {
        const auto t0 = now();
        const int rc = batches[g]->sync();
        t_wait += ms(t0, now());
        return rc;
    };
    auto finish_read = [&](size_t g) -> int {
        const size_t lo = glo[g], hi = glo[g + 1];
        jb::Batch *b = batches[g].get();
        int rc = JB_OK;
        const auto t0 = now();
        for (size_t u = lo; u < hi; u++) {
            const size_t ns = (size_t)b->T[u - lo] * b->voice.fperiod;
            n_samples[u] = ns;
            if (!ns)
                continue;
            // 2 MB alignment + MADV_HUGEPAGE: where transparent huge pages are allowed, the first touch of
            // the buffer (by the scatter threads) takes 512x fewer faults; free() releases it as usual
            const size_t bytes = ns * elem;
            if (bytes >= (4u << 20) && posix_memalign(&pcm[u], 2u << 20, bytes) == 0)
                madvise(pcm[u], bytes, MADV_HUGEPAGE);
            else
                pcm[u] = malloc(bytes);
            if (!pcm[u]) {
                jb::set_error("out of host memory");
                return JB_ERR_INVALID;
            }
        }
        // whole slab through the pinned staging ring into the per-utterance buffers
        rc = b->read_pcm_split(pcm + lo, elem);
        batches[g].reset(); // device memory and streams back to the pools
        t_d2h += ms(t0, now());
        return rc;
    };
    auto finish = [&](size_t g) -> int {
        const int rc = finish_sync(g);
        return rc ? rc : finish_read(g);
    };
    // f64 in groups (JB_SYNTH_GROUPS only): all groups' GPU work first, then the read-backs, so that no
    // read-back runs beside kernels
    const bool d2h_last = elem == 8;

    int rc = JB_OK;
    if (ngroups == 1) {
        if (!(rc = front(0, n_utts)) && !(rc = launch(0)))
            rc = finish(0);
    } else {
        // finisher thread: groups in order, as soon as they are launched
        std::mutex mu;
        std::condition_variable cv;
        size_t launched = 0;
        bool stop = false;
        int frc = JB_OK;
        std::string ferr;
        std::thread finisher([&]() {
            for (size_t g = 0; g < ngroups; g++) {
                {
                    std::unique_lock<std::mutex> lk(mu);
                    cv.wait(lk, [&] { return launched > g || stop; });
                    if (launched <= g)
                        return;
                }
                const int r = d2h_last ? finish_sync(g) : finish(g);
                if (r) {
                    std::lock_guard<std::mutex> lk(mu);
                    frc = r;
                    ferr = jb::g_err;
                    return;
                }
            }
            for (size_t g = 0; d2h_last && g < ngroups; g++) {
                const int r = finish_read(g);
                if (r) {
                    std::lock_guard<std::mutex> lk(mu);
                    frc = r;
                    ferr = jb::g_err;
                    return;
                }
            }
        });
        for (size_t g = 0; g < ngroups && !rc; g++) {
            if ((rc = front(glo[g], glo[g + 1])) || (rc = launch(g)))
                break;
            std::lock_guard<std::mutex> lk(mu);
            launched = g + 1;
            cv.notify_all();
            if (frc)
                break;
        }
        {
            std::lock_guard<std::mutex> lk(mu);
            stop = true;
            cv.notify_all();
        }
        finisher.join();
        if (!rc && frc) {
            rc = frc;
            jb::set_error(ferr);
        }
    }
    if (rc) {
        batches.clear();
        for (size_t u = 0; u < n_utts; u++) {
            free(pcm[u]);
            pcm[u] = nullptr;
            n_samples[u] = 0;
        }
        return rc;
    }
    if (timing)
        fprintf(stderr,
                "jb_synthesize_batch: %zu group(s), wall %.1f ms; front half %.1f ms, upload+create+enqueue %.1f ms, "
                "wait for the GPU %.1f ms, D2H %.1f ms\n",
                ngroups, ms(t_start, now()), t_front, t_create, t_wait, t_d2h);
    return JB_OK;
}

extern "C" {

int jb_synthesize_batch(const jb_engine *e, const char *const *lines, const size_t *line_off,
                        size_t n_utts, int32_t device, double **pcm, size_t *n_samples)
{
    return jb::synthesize_batch_impl(e, lines, line_off, n_utts, device, sizeof(double), (void **)pcm, n_samples);
}

int jb_synthesize_batch_i16(const jb_engine *e, const char *const *lines, const size_t *line_off,
                            size_t n_utts, int32_t device, int16_t **pcm, size_t *n_samples)
{
    return jb::synthesize_batch_impl(e, lines, line_off, n_utts, device, sizeof(int16_t), (void **)pcm, n_samples);
}

void jb_pcm_i16_free(int16_t *p) { free(p); }

int jb_synthesize(const jb_engine *e, const char *const *lines, size_t n, double **pcm, size_t *n_samples)
{
    if (!pcm || !n_samples)
        return JB_ERR_INVALID;
    size_t off[2] = {0, n};
    return jb_synthesize_batch(e, lines, off, 1, -1, pcm, n_samples);
}

// ---- generator (src/speech.rs) ----
int jb_generator_new(const jb_engine *e, const char *const *lines, size_t n, jb_generator **out)
{
    if (!e || !out)
        return JB_ERR_INVALID;
    *out = nullptr;
    jb::States st;
    int rc = build_states(*CENG(e), lines, n, st, false, host_threads_default());
    if (rc)
        return rc;
    std::unique_ptr<jb::Generator> g(new jb::Generator());
    jb_batch_opts opts{};
    opts.device = -1;
    opts.flags = CENG(e)->cond.batch_invariant ? (JB_BATCH_SERIAL | JB_BATCH_SERIAL_GV) : 0;
    if (const char *ev = getenv("JB_GENERATOR_TEST_GANG_TIMEOUT")) // test aid, as JB_BATCH_TEST_GANG_TIMEOUT
        if (atoi(ev) != 0)
            opts.flags |= JB_BATCH_TEST_GANG_TIMEOUT;
    jb::Batch *b = nullptr;
    if ((rc = jb::Batch::create(&CENG(e)->desc, &st.utt, 1, &opts, &b)))
        return rc;
    g->batch.reset(b);
    g->fperiod = b->voice.fperiod;
    g->total = b->T[0];
    // Engine::generator runs all three MLPGs before returning (src/engine.rs:333-357); here they are
    // enqueued, with the vocoder behind them, and the call returns while the device works
    if ((rc = b->build_generator_work()) || (rc = b->run(false)))
        return rc;
    *out = (jb_generator *)g.release();
    return JB_OK;
}

// SpeechGenerator::new(fperiod, vocoder, spectrum, lf0, lpf) on tracks the caller holds (src/speech.rs:25-50),
// to be stepped with jb_generator_step (generate_step, :65-82): the same three panics as error codes
int jb_generator_new_from_tracks(const jb_voice_desc *voice, const jb_track_utt *utt, const jb_batch_opts *opts,
                                 jb_generator **out)
{
    if (!voice || !utt || !out)
        return JB_ERR_INVALID;
    *out = nullptr;
    std::unique_ptr<jb::Generator> g(new jb::Generator());
    jb_batch_opts o{};
    if (opts)
        o = *opts;
    else
        o.device = -1;
    o.flags &= ~(uint32_t)(JB_BATCH_PCM_I16 | JB_BATCH_MLPG_ONLY); // generate_step hands out f64 samples
    jb::Batch *b = nullptr;
    jb::TrackSrc src{utt};
    int rc = jb::Batch::create(voice, nullptr, 1, &o, &b, nullptr, &src);
    if (rc)
        return rc;
    g->batch.reset(b);
    g->fperiod = b->voice.fperiod;
    g->total = b->T[0];
    if ((rc = b->build_generator_work()) || (rc = b->run(false)))
        return rc;
    *out = (jb_generator *)g.release();
    return JB_OK;
}

size_t jb_generator_fperiod(const jb_generator *g) { return g ? ((const jb::Generator *)g)->fperiod : 0; }
size_t jb_generator_synthesized_frames(const jb_generator *g)
{
    return g ? ((const jb::Generator *)g)->next : 0;
}
size_t jb_generator_total_frames(const jb_generator *g) { return g ? ((const jb::Generator *)g)->total : 0; }

// waits for the whole-utterance run (and its certification / redo) once
static int generator_finish(jb::Generator *g)
{
    if (g->ahead_ready)
        return JB_OK;
    int rc = g->batch->sync();
    if (rc)
        return rc;
    g->ahead_ready = true;
    return JB_OK;
}

long jb_generator_step(jb_generator *hg, double *buf, size_t buf_len)
{
    jb::Generator *g = (jb::Generator *)hg;
    if (!g)
        return JB_ERR_INVALID;
    if (g->total <= g->next)
        return 0;
    if (buf_len < g->fperiod || !buf) {
        set_error("The length of speech buffer must be larger than fperiod.");
        return JB_ERR_BUFFER;
    }
    jb::Batch *b = g->batch.get();
    if (hipSetDevice(b->device) != hipSuccess)
        return JB_ERR_DEVICE;
    int rc;
    if (!g->ahead_ready && (g->next >= kGenSerialFrames || hipEventQuery(b->ev_voc_done) == hipSuccess) &&
        (rc = generator_finish(g)))
        return rc;
    if (!g->ahead_ready) {
        // the utterance is still in flight: this frame from the serial recursion on the side stream
        hipStream_t ss = b->stream_lf0;
        hipError_t he;
        if (!g->serial_armed) {
            // Parameter generation must be complete before a frame is served -- and it is not if the resident GV
            // kernel gave up in formation (possible without a fault when several such launches share a device:
            // generators made back to back): the flag is read here, once, not only in Batch::sync, which would
            // redo the step after the head of the utterance had gone out computed from an unfinished MCP track.
            if ((he = hipEventSynchronize(b->ev_mlpg_done)) != hipSuccess)
                return hip_fail(he, "generator: parameter generation");
            bool timed_out = false;
            if ((rc = b->gang_timeout_seen(&timed_out)))
                return rc;
            if (timed_out) {
                if ((rc = generator_finish(g))) // sync(): the step again with the multi-launch GV, certified
                    return rc;
            } else {
                g->serial_armed = true;
            }
        }
        if (g->serial_armed && !g->ahead_ready) {
            // the serial recursion writes the frame into a buffer of its own (the whole-utterance run is writing
            // the same frames of the PCM slab, and from the throughput kernel not bit for bit the same values)
            jb::VocDev vdg = b->vd;
            vdg.pcm = b->gen_pcm - g->next * g->fperiod; // utterance 0, frame `next` -> gen_pcm[0 .. fperiod)
            if ((he = launch_vocoder(b->bd, vdg, b->gen_work_dev + g->next, 1, ss)) != hipSuccess)
                return hip_fail(he, "k_vocoder");
            if ((he = hipStreamSynchronize(ss)) != hipSuccess)
                return hip_fail(he, "generator step");
            if ((rc = b->read(b->gen_pcm, buf, g->fperiod * sizeof(double), false)))
                return rc;
            g->next++;
            return (long)g->fperiod;
        }
    }
    if (g->next < g->cache_first || g->next >= g->cache_first + g->cache_frames) {
        const size_t nf = std::min(kGenBlockFrames, g->total - g->next);
        g->cache.resize(nf * g->fperiod);
        if ((rc = b->read(b->vd.pcm + g->next * g->fperiod, g->cache.data(), nf * g->fperiod * sizeof(double), false)))
            return rc;
        g->cache_first = g->next;
        g->cache_frames = nf;
    }
    memcpy(buf, g->cache.data() + (g->next - g->cache_first) * g->fperiod, g->fperiod * sizeof(double));
    g->next++;
    return (long)g->fperiod;
}

long jb_generator_step_n(jb_generator *hg, double *buf, size_t buf_len, size_t max_frames)
{
    jb::Generator *g = (jb::Generator *)hg;
    if (!g)
        return JB_ERR_INVALID;
    if (g->total <= g->next || max_frames == 0)
        return 0;
    if (buf_len < g->fperiod || !buf) {
        set_error("The length of speech buffer must be larger than fperiod.");
        return JB_ERR_BUFFER;
    }
    const size_t nf = std::min(std::min(max_frames, g->total - g->next), buf_len / g->fperiod);
    jb::Batch *b = g->batch.get();
    if (hipSetDevice(b->device) != hipSuccess)
        return JB_ERR_DEVICE;
    int rc = generator_finish(g);
    if (rc)
        return rc;
    if ((rc = b->read(b->vd.pcm + g->next * g->fperiod, buf, nf * g->fperiod * sizeof(double), false)))
        return rc;
    g->next += nf;
    return (long)(nf * g->fperiod);
}

void jb_generator_free(jb_generator *g) { delete (jb::Generator *)g; }

} // extern "C"
