// jb_voice.cpp -- .htsvoice reader and decision-tree search (host, cold path).
//
// Format facts follow /root/reference/src/model/parser:
//   sections [GLOBAL]/[STREAM]/[POSITION]/[DATA]      mod.rs:76-102
//   POSITION ranges are inclusive offsets into DATA    mod.rs:176-187
//   pdf blob = ntree x u32 npdf, then npdf x pdf_len f32 (little endian)  model/mod.rs:38-60
//   tree rows "id question NO-child YES-child"         model/tree.rs:85-107
//   leaf name -> trailing digit run = 1-based pdf idx  model/tree.rs:58-84
// Question matching is glob matching over the label string: the reference defers
// to jlabel-question 0.1.10 (src/model/voice/question.rs), whose patterns are the
// HTS question-set globs stored in the voice file itself.
#include "jb_voice.h"

#include <cctype>
#include <charconv>
#include <cstring>
#include <fstream>
#include <map>
#include <sstream>
#include <unordered_map>

namespace jb {

bool glob_match(std::string_view pat, std::string_view s)
{
    size_t p = 0, i = 0, star = std::string_view::npos, mark = 0;
    while (i < s.size()) {
        if (p < pat.size() && (pat[p] == '?' || (pat[p] != '*' && pat[p] == s[i]))) {
            p++;
            i++;
        } else if (p < pat.size() && pat[p] == '*') {
            star = p++;
            mark = i;
        } else if (star != std::string_view::npos) {
            p = star + 1;
            i = ++mark;
        } else {
            return false;
        }
    }
    while (p < pat.size() && pat[p] == '*')
        p++;
    return p == pat.size();
}

void Question::compile()
{
    compiled.clear();
    for (const std::string &p : patterns) {
        const size_t n = p.size();
        const bool meta_inside = n > 2 && p.find_first_of("*?", 1) < n - 1;
        const bool has_q = p.find('?') != std::string::npos;
        const bool ls = n > 0 && p.front() == '*', ts = n > 0 && p.back() == '*';
        Kind k = Glob;
        std::string core;
        if (has_q || meta_inside || n == 0)
            k = Glob;
        else if (n == 1 && ls)
            k = Any;
        else if (ls && ts && n >= 2) {
            k = n == 2 ? Any : Contains; // "**" matches everything
            core = p.substr(1, n - 2);
        } else if (ts) {
            k = Prefix;
            core = p.substr(0, n - 1);
        } else if (ls) {
            k = Suffix;
            core = p.substr(1);
        } else {
            k = Exact;
            core = p;
        }
        compiled.emplace_back(k, k == Glob ? p : core);
    }
}

bool Question::test(std::string_view label) const
{
    if (compiled.size() != patterns.size()) { // not compiled (hand-built question): general matcher
        for (const auto &p : patterns)
            if (glob_match(p, label))
                return true;
        return false;
    }
    for (const auto &[k, t] : compiled) {
        bool hit;
        switch (k) {
        case Any: hit = true; break;
        case Contains: hit = label.find(t) != std::string_view::npos; break;
        case Prefix: hit = label.size() >= t.size() && label.compare(0, t.size(), t) == 0; break;
        case Suffix:
            hit = label.size() >= t.size() && label.compare(label.size() - t.size(), t.size(), t) == 0;
            break;
        case Exact: hit = label == t; break;
        default: hit = glob_match(t, label); break;
        }
        if (hit)
            return true;
    }
    return false;
}

std::vector<int8_t> *QuestionMemo::of(const Model *m, size_t nq)
{
    for (Slot &s : slots)
        if (s.m == m)
            return &s.v;
    slots.push_back(Slot{m, std::vector<int8_t>(nq, (int8_t)-1)});
    return &slots.back().v;
}

void QuestionMemo::reset()
{
    for (Slot &s : slots)
        std::fill(s.v.begin(), s.v.end(), (int8_t)-1);
}

int Tree::search(const std::vector<Question> &qs, std::string_view label, std::vector<int8_t> *memo) const
{
    if (nodes.empty())
        return single_leaf;
    int32_t i = 0;
    for (;;) {
        const TreeNode &n = nodes[(size_t)i];
        bool yes;
        if (memo) {
            int8_t &c = (*memo)[(size_t)n.question];
            if (c < 0)
                c = qs[(size_t)n.question].test(label) ? 1 : 0;
            yes = c != 0;
        } else {
            yes = qs[(size_t)n.question].test(label);
        }
        int32_t next = yes ? n.yes : n.no;
        if (next < 0)
            return -next;
        i = next;
    }
}

void Model::get_index(int state_index, std::string_view label, int &tree_pos, int &pdf_index,
                      QuestionMemo *memo) const
{
    tree_pos = -1;
    for (size_t i = 0; i < trees.size(); i++)
        if (trees[i].state == state_index) {
            tree_pos = (int)i;
            break;
        }
    const Tree &t = trees[tree_pos < 0 ? 0 : (size_t)tree_pos];
    pdf_index = t.search(questions, label, memo ? memo->of(this, questions.size()) : nullptr);
}

const float *Model::get_parameter(int state_index, std::string_view label, QuestionMemo *memo) const
{
    int tp, pi;
    get_index(state_index, label, tp, pi, memo);
    if (tp < 0 || pi < 1 || pi > npdf[(size_t)tp])
        throw ModelError("index not found"); // reference: todo!() (voice/model.rs:76-79)
    return pdf[(size_t)tp].data() + (size_t)(pi - 1) * (size_t)pdf_len;
}

bool GlobalMeta::operator==(const GlobalMeta &o) const
{
    return hts_voice_version == o.hts_voice_version && sampling_frequency == o.sampling_frequency &&
           frame_period == o.frame_period && num_states == o.num_states &&
           num_streams == o.num_streams && stream_type == o.stream_type &&
           fullcontext_format == o.fullcontext_format &&
           fullcontext_version == o.fullcontext_version && gv_off_patterns == o.gv_off_patterns;
}

// ---------------------------------------------------------------------------
namespace {

using KV = std::map<std::string, std::string>;

KV parse_kv(std::string_view sec)
{
    KV kv;
    size_t p = 0;
    while (p < sec.size()) {
        size_t e = sec.find('\n', p);
        if (e == std::string_view::npos)
            e = sec.size();
        std::string_view line = sec.substr(p, e - p);
        size_t c = line.find(':');
        if (c != std::string_view::npos)
            kv[std::string(line.substr(0, c))] = std::string(line.substr(c + 1));
        p = e + 1;
    }
    return kv;
}

const std::string &need(const KV &kv, const std::string &k)
{
    auto it = kv.find(k);
    if (it == kv.end())
        throw ModelError("missing header key " + k);
    return it->second;
}

int to_int(const std::string &s, const char *what)
{
    int v = 0;
    auto r = std::from_chars(s.data(), s.data() + s.size(), v);
    if (r.ec != std::errc() || r.ptr != s.data() + s.size())
        throw ModelError(std::string("bad integer for ") + what + ": " + s);
    return v;
}

std::vector<std::string> split(const std::string &s, char d)
{
    std::vector<std::string> out;
    if (s.empty())
        return out;
    size_t p = 0;
    for (;;) {
        size_t e = s.find(d, p);
        out.push_back(s.substr(p, e == std::string::npos ? std::string::npos : e - p));
        if (e == std::string::npos)
            break;
        p = e + 1;
    }
    return out;
}

std::vector<std::string> quoted_list(std::string_view s)
{
    std::vector<std::string> out;
    size_t p = 0;
    while ((p = s.find('"', p)) != std::string_view::npos) {
        size_t e = s.find('"', p + 1);
        if (e == std::string_view::npos)
            throw ModelError("unterminated pattern");
        out.emplace_back(s.substr(p + 1, e - p - 1));
        p = e + 1;
    }
    return out;
}

struct Range {
    size_t a, b;
};

Range parse_range(const std::string &s)
{
    size_t d = s.find('-');
    if (d == std::string::npos)
        throw ModelError("bad range " + s);
    Range r{(size_t)std::stoull(s.substr(0, d)), (size_t)std::stoull(s.substr(d + 1))};
    if (r.b < r.a)
        throw ModelError("bad range " + s);
    return r;
}

struct RawChild {
    bool is_node;
    long v;
};

RawChild parse_child(std::string_view t)
{
    if (t.size() >= 2 && t.front() == '"' && t.back() == '"')
        t = t.substr(1, t.size() - 2);
    size_t i = (!t.empty() && t[0] == '-') ? 1 : 0;
    bool num = t.size() > i;
    for (size_t k = i; k < t.size(); k++)
        num = num && std::isdigit((unsigned char)t[k]);
    if (num)
        return {true, std::stol(std::string(t))};
    size_t e = t.size();
    while (e > 0 && std::isdigit((unsigned char)t[e - 1]))
        e--;
    if (e == t.size())
        throw ModelError("leaf without pdf index: " + std::string(t));
    return {false, std::stol(std::string(t.substr(e)))};
}

struct Tok {
    std::string_view s;
    size_t p = 0;
    void ws()
    {
        while (p < s.size() && std::isspace((unsigned char)s[p]))
            p++;
    }
    bool eof()
    {
        ws();
        return p >= s.size();
    }
    std::string_view word()
    {
        ws();
        size_t b = p;
        while (p < s.size() && !std::isspace((unsigned char)s[p]))
            p++;
        return s.substr(b, p - b);
    }
    char peek()
    {
        ws();
        return p < s.size() ? s[p] : 0;
    }
};

void parse_trees(Model &m, std::string_view text)
{
    std::unordered_map<std::string, int32_t> qidx;
    Tok tk{text};
    while (!tk.eof()) {
        if (tk.s.substr(tk.p, 2) == "QS") {
            tk.p += 2;
            std::string name(tk.word());
            size_t lb = tk.s.find('{', tk.p), rb = tk.s.find('}', tk.p);
            if (lb == std::string_view::npos || rb == std::string_view::npos || rb < lb)
                throw ModelError("bad QS row");
            Question q;
            q.patterns = quoted_list(tk.s.substr(lb + 1, rb - lb - 1));
            q.compile();
            qidx[name] = (int32_t)m.questions.size();
            m.questions.push_back(std::move(q));
            tk.p = rb + 1;
        } else if (tk.s.substr(tk.p, 3) == "{*}") {
            tk.p += 3;
            if (tk.peek() != '[')
                throw ModelError("tree without state");
            size_t rb = tk.s.find(']', tk.p);
            Tree t;
            t.state = to_int(std::string(tk.s.substr(tk.p + 1, rb - tk.p - 1)), "tree state");
            tk.p = rb + 1;
            if (tk.peek() != '{') {
                RawChild c = parse_child(tk.word());
                if (c.is_node)
                    throw ModelError("single-leaf tree expected");
                t.single_leaf = (int)c.v;
                m.trees.push_back(std::move(t));
                continue;
            }
            tk.p++;
            struct Raw {
                long id;
                int32_t q;
                RawChild no, yes;
            };
            std::vector<Raw> raw;
            while (tk.peek() != '}') {
                if (tk.eof())
                    throw ModelError("unterminated tree");
                Raw r;
                r.id = std::stol(std::string(tk.word()));
                std::string qn(tk.word());
                auto it = qidx.find(qn);
                if (it == qidx.end())
                    throw ModelError("unknown question " + qn);
                r.q = it->second;
                r.no = parse_child(tk.word()); // first child column = NO branch
                r.yes = parse_child(tk.word());
                raw.push_back(r);
            }
            tk.p++;
            std::unordered_map<long, int32_t> pos;
            for (size_t i = 0; i < raw.size(); i++)
                pos[raw[i].id] = (int32_t)i;
            auto resolve = [&](const RawChild &c) -> int32_t {
                if (!c.is_node)
                    return (int32_t)-c.v;
                auto it = pos.find(c.v);
                if (it == pos.end())
                    throw ModelError("dangling tree node");
                return it->second;
            };
            for (const Raw &r : raw)
                t.nodes.push_back({r.q, resolve(r.yes), resolve(r.no)});
            m.trees.push_back(std::move(t));
        } else {
            throw ModelError("unexpected token in tree section");
        }
    }
}

Model parse_model(std::string_view data, Range tree, Range pdf, int pdf_len)
{
    if (tree.b >= data.size() || pdf.b >= data.size() || tree.a > tree.b || pdf.a > pdf.b)
        throw ModelError("position out of range");
    if (pdf_len <= 0)
        throw ModelError("pdf length out of range");
    Model m;
    m.pdf_len = pdf_len;
    parse_trees(m, data.substr(tree.a, tree.b - tree.a + 1));
    // (counts and lengths come from the file: every size is checked against the bytes that are LEFT, never by
    // forming a pointer past them -- a count of 0xffffffff times a vector length wraps a pointer sum)
    const uint8_t *p = (const uint8_t *)data.data() + pdf.a;
    const uint8_t *end = (const uint8_t *)data.data() + pdf.b + 1;
    auto left = [&]() { return (size_t)(end - p); };
    auto rd32 = [&](const uint8_t *q) {
        return (uint32_t)q[0] | (uint32_t)q[1] << 8 | (uint32_t)q[2] << 16 | (uint32_t)q[3] << 24;
    };
    for (size_t k = 0; k < m.trees.size(); k++) {
        if (left() < 4)
            throw ModelError("pdf blob truncated");
        const uint32_t np = rd32(p);
        if (np > 0x7fffffffu)
            throw ModelError("pdf count out of range");
        m.npdf.push_back((int)np);
        p += 4;
    }
    for (size_t k = 0; k < m.trees.size(); k++) {
        if ((size_t)m.npdf[k] > left() / 4 / (size_t)pdf_len)
            throw ModelError("pdf blob truncated");
        size_t cnt = (size_t)m.npdf[k] * (size_t)pdf_len;
        std::vector<float> v(cnt);
        for (size_t i = 0; i < cnt; i++) {
            uint32_t u = rd32(p + 4 * i);
            float f;
            std::memcpy(&f, &u, 4);
            v[i] = f;
        }
        p += 4 * cnt;
        m.pdf.push_back(std::move(v));
    }
    if (p != end)
        throw ModelError("pdf blob has trailing bytes");
    return m;
}

} // namespace

std::shared_ptr<Voice> parse_htsvoice(const uint8_t *bytes, size_t n)
{
    std::string_view all((const char *)bytes, n);
    auto find_sec = [&](std::string_view tag, size_t from) {
        size_t p = from;
        for (;;) {
            p = all.find(tag, p);
            if (p == std::string_view::npos)
                throw ModelError("section " + std::string(tag) + " not found");
            if (p == 0 || all[p - 1] == '\n')
                return p;
            p++;
        }
    };
    size_t g = find_sec("[GLOBAL]\n", 0);
    size_t s = find_sec("[STREAM]\n", g);
    size_t po = find_sec("[POSITION]\n", s);
    size_t d = find_sec("[DATA]\n", po);
    KV G = parse_kv(all.substr(g + 9, s - g - 9));
    KV S = parse_kv(all.substr(s + 9, po - s - 9));
    KV P = parse_kv(all.substr(po + 11, d - po - 11));
    std::string_view data = all.substr(d + 7);

    auto v = std::make_shared<Voice>();
    GlobalMeta &m = v->meta;
    m.hts_voice_version = need(G, "HTS_VOICE_VERSION");
    m.sampling_frequency = to_int(need(G, "SAMPLING_FREQUENCY"), "SAMPLING_FREQUENCY");
    m.frame_period = to_int(need(G, "FRAME_PERIOD"), "FRAME_PERIOD");
    m.num_states = to_int(need(G, "NUM_STATES"), "NUM_STATES");
    m.num_streams = to_int(need(G, "NUM_STREAMS"), "NUM_STREAMS");
    m.stream_type = split(need(G, "STREAM_TYPE"), ',');
    m.fullcontext_format = need(G, "FULLCONTEXT_FORMAT");
    m.fullcontext_version = need(G, "FULLCONTEXT_VERSION");
    if (auto it = G.find("GV_OFF_CONTEXT"); it != G.end())
        m.gv_off_patterns = quoted_list(it->second);
    v->gv_off.patterns = m.gv_off_patterns;
    v->gv_off.compile();
    if ((int)m.stream_type.size() != m.num_streams || m.num_states <= 0 || m.num_states > (1 << 20))
        throw ModelError("inconsistent global header");

    v->duration = parse_model(data, parse_range(need(P, "DURATION_TREE")),
                              parse_range(need(P, "DURATION_PDF")), m.num_states * 2);
    for (const std::string &nm : m.stream_type) {
        StreamModel sm;
        sm.name = nm;
        auto key = [&](const char *k) { return std::string(k) + "[" + nm + "]"; };
        sm.vector_length = to_int(need(S, key("VECTOR_LENGTH")), "VECTOR_LENGTH");
        sm.num_windows = to_int(need(S, key("NUM_WINDOWS")), "NUM_WINDOWS");
        // (pdf_len below is a product of the two: keep it an int)
        if (sm.vector_length < 0 || sm.vector_length > (1 << 20) || sm.num_windows < 0 || sm.num_windows > 255)
            throw ModelError("VECTOR_LENGTH / NUM_WINDOWS out of range");
        sm.is_msd = to_int(need(S, key("IS_MSD")), "IS_MSD") != 0;
        sm.use_gv = to_int(need(S, key("USE_GV")), "USE_GV") != 0;
        if (auto it = S.find(key("OPTION")); it != S.end())
            sm.options = split(it->second, ',');
        for (const std::string &r : split(need(P, key("STREAM_WIN")), ',')) {
            Range wr = parse_range(r);
            if (wr.b >= data.size())
                throw ModelError("window out of range");
            std::istringstream is(std::string(data.substr(wr.a, wr.b - wr.a + 1)));
            size_t cnt;
            if (!(is >> cnt) || cnt > wr.b - wr.a + 1) // (a coefficient takes at least one byte of the row's text)
                throw ModelError("bad window row");
            std::vector<double> w(cnt);
            for (double &c : w)
                if (!(is >> c))
                    throw ModelError("bad window row");
            sm.windows.push_back(std::move(w));
        }
        if ((int)sm.windows.size() != sm.num_windows)
            throw ModelError("window count mismatch");
        sm.stream = parse_model(data, parse_range(need(P, key("STREAM_TREE"))),
                                parse_range(need(P, key("STREAM_PDF"))),
                                sm.vector_length * sm.num_windows * 2 + (sm.is_msd ? 1 : 0));
        if (sm.use_gv) {
            auto t = P.find(key("GV_TREE")), p = P.find(key("GV_PDF"));
            if (t == P.end() || p == P.end())
                throw ModelError("USE_GV is true, but positions for GV is not set");
            sm.gv = parse_model(data, parse_range(t->second), parse_range(p->second),
                                sm.vector_length * 2);
        }
        v->streams.push_back(std::move(sm));
    }
    return v;
}

std::shared_ptr<Voice> load_htsvoice(const std::string &path)
{
    std::ifstream f(path, std::ios::binary);
    if (!f)
        throw ModelError("Io failed: cannot open " + path);
    std::vector<uint8_t> b((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    return parse_htsvoice(b.data(), b.size());
}

} // namespace jb
