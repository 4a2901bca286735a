/*
 * oracle/jbo_voice.c -- CPU ORACLE (test infrastructure, NOT the product).
 *
 * .htsvoice reader + decision-tree search, restating (cold path, needed only to
 * reach the reference's golden samples from label text):
 *   /root/reference/src/model/parser/mod.rs:58-187      (sections, data section)
 *   /root/reference/src/model/parser/model/mod.rs:17-136 (pdf blob, tree conversion)
 *   /root/reference/src/model/parser/model/tree.rs:40-107 (node rows: id question NO YES)
 *   /root/reference/src/model/parser/model/question.rs:44-84 (QS rows)
 *   /root/reference/src/model/voice/model.rs:51-82      (get_index / get_parameter)
 *   /root/reference/src/model/voice/tree.rs:14-27       (search_node)
 * Question matching: the reference defers to the un-vendored crate
 * jlabel-question 0.1.10 (call sites src/model/voice/question.rs:11-23); its
 * published behaviour for HTS question sets is glob matching (`*`, `?`) of each
 * pattern against the full-context label string, which is what is restated
 * here and pinned by src/model/mod.rs:191-212 (tree indices) and :234-392 (pdfs).
 */
#include "jbo_internal.h"

#include <ctype.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* ---- glob ------------------------------------------------------------- */
int jbo_glob(const char *pat, const char *s)
{
    const char *star = NULL, *ss = NULL;
    while (*s) {
        if (*pat == '?' || (*pat != '*' && *pat == *s)) {
            pat++;
            s++;
        } else if (*pat == '*') {
            star = pat++;
            ss = s;
        } else if (star) {
            pat = star + 1;
            s = ++ss;
        } else {
            return 0;
        }
    }
    while (*pat == '*')
        pat++;
    return *pat == 0;
}

static int question_test(const jbo_question *q, const char *label)
{
    for (int i = 0; i < q->npat; i++)
        if (jbo_glob(q->pats[i], label))
            return 1;
    return 0;
}

/* ---- header ------------------------------------------------------------ */
static char *dupn(const char *s, size_t n)
{
    char *r = (char *)malloc(n + 1);
    memcpy(r, s, n);
    r[n] = 0;
    return r;
}

/* find "KEY:" at a line start inside [beg,end); returns malloc'd value */
static char *header_get(const char *beg, const char *end, const char *key)
{
    size_t kl = strlen(key);
    const char *p = beg;
    while (p < end) {
        const char *eol = memchr(p, '\n', (size_t)(end - p));
        if (!eol)
            eol = end;
        if ((size_t)(eol - p) > kl && memcmp(p, key, kl) == 0 && p[kl] == ':')
            return dupn(p + kl + 1, (size_t)(eol - p) - kl - 1);
        p = eol + 1;
    }
    return NULL;
}

static int parse_range(const char *s, long *a, long *b)
{
    return sscanf(s, "%ld-%ld", a, b) == 2 ? 0 : -1;
}

/* ---- tree text --------------------------------------------------------- */
typedef struct {
    const char *p, *end;
} cur_t;

static void skip_ws(cur_t *c)
{
    while (c->p < c->end && (*c->p == ' ' || *c->p == '\n' || *c->p == '\t' || *c->p == '\r'))
        c->p++;
}

/* token: run of non-space chars (quotes kept) */
static int next_tok(cur_t *c, const char **tb, size_t *tl)
{
    skip_ws(c);
    if (c->p >= c->end)
        return 0;
    *tb = c->p;
    while (c->p < c->end && !(*c->p == ' ' || *c->p == '\n' || *c->p == '\t' || *c->p == '\r'))
        c->p++;
    *tl = (size_t)(c->p - *tb);
    return 1;
}

/* child token -> node id (is_node=1) or pdf index (trailing digit run) */
static int parse_child(const char *t, size_t n, int *is_node, long *val)
{
    if (n >= 2 && t[0] == '"' && t[n - 1] == '"') {
        t++;
        n -= 2;
    }
    size_t i = 0;
    if (n > 0 && t[0] == '-')
        i = 1;
    int alldig = (n > i);
    for (size_t k = i; k < n; k++)
        if (!isdigit((unsigned char)t[k]))
            alldig = 0;
    if (alldig) {
        *is_node = 1;
        *val = strtol(dupn(t, n), NULL, 10);
        return 0;
    }
    /* trailing digit run (src/model/parser/model/tree.rs:60-77) */
    size_t e = n;
    while (e > 0 && isdigit((unsigned char)t[e - 1]))
        e--;
    if (e == n)
        return -1;
    char *d = dupn(t + e, n - e);
    *is_node = 0;
    *val = strtol(d, NULL, 10);
    free(d);
    return 0;
}

static int find_question(const jbo_model *m, const char *name, size_t n)
{
    for (int i = 0; i < m->nq; i++)
        if (strlen(m->qs[i].name) == n && memcmp(m->qs[i].name, name, n) == 0)
            return i;
    return -1;
}

static int parse_tree_text(jbo_model *m, const char *beg, const char *end)
{
    cur_t c = {beg, end};
    const char *tb;
    size_t tl;
    int qcap = 0, tcap = 0;
    m->nq = 0;
    m->qs = NULL;
    m->ntree = 0;
    m->trees = NULL;
    for (;;) {
        const char *save = c.p;
        if (!next_tok(&c, &tb, &tl))
            break;
        if (tl == 2 && memcmp(tb, "QS", 2) == 0) {
            if (m->nq == qcap) {
                qcap = qcap ? qcap * 2 : 64;
                m->qs = (jbo_question *)realloc(m->qs, sizeof(jbo_question) * (size_t)qcap);
            }
            jbo_question *q = &m->qs[m->nq++];
            next_tok(&c, &tb, &tl);
            q->name = dupn(tb, tl);
            q->npat = 0;
            q->pats = NULL;
            /* { "a","b" } -- patterns may not contain spaces in HTS voices, but
             * parse by quotes to be safe */
            while (c.p < c.end && *c.p != '{')
                c.p++;
            c.p++;
            int pcap = 0;
            while (c.p < c.end && *c.p != '}') {
                if (*c.p == '"') {
                    const char *s = ++c.p;
                    while (c.p < c.end && *c.p != '"')
                        c.p++;
                    if (q->npat == pcap) {
                        pcap = pcap ? pcap * 2 : 8;
                        q->pats = (char **)realloc(q->pats, sizeof(char *) * (size_t)pcap);
                    }
                    q->pats[q->npat++] = dupn(s, (size_t)(c.p - s));
                }
                c.p++;
            }
            c.p++;
        } else if (tl >= 3 && memcmp(tb, "{*}", 3) == 0) {
            /* "{*}[2]" possibly split by spaces */
            c.p = save;
            skip_ws(&c);
            c.p += 3;
            skip_ws(&c);
            if (*c.p != '[')
                return -1;
            c.p++;
            long st = strtol(c.p, (char **)&c.p, 10);
            if (*c.p != ']')
                return -1;
            c.p++;
            if (m->ntree == tcap) {
                tcap = tcap ? tcap * 2 : 8;
                m->trees = (jbo_tree *)realloc(m->trees, sizeof(jbo_tree) * (size_t)tcap);
            }
            jbo_tree *t = &m->trees[m->ntree++];
            memset(t, 0, sizeof *t);
            t->state = (int)st;
            skip_ws(&c);
            if (*c.p != '{') {
                /* single leaf */
                next_tok(&c, &tb, &tl);
                int isn;
                long v;
                if (parse_child(tb, tl, &isn, &v) || isn)
                    return -1;
                t->single_leaf = (int)v;
                continue;
            }
            c.p++;
            int ncap = 0;
            for (;;) {
                skip_ws(&c);
                if (*c.p == '}') {
                    c.p++;
                    break;
                }
                if (t->nnode == ncap) {
                    ncap = ncap ? ncap * 2 : 64;
                    t->nodes = (jbo_node *)realloc(t->nodes, sizeof(jbo_node) * (size_t)ncap);
                }
                jbo_node *nd = &t->nodes[t->nnode++];
                next_tok(&c, &tb, &tl);
                char *ids = dupn(tb, tl);
                nd->id = strtol(ids, NULL, 10);
                free(ids);
                next_tok(&c, &tb, &tl);
                nd->q = find_question(m, tb, tl);
                if (nd->q < 0)
                    return -1;
                /* first child column is the NO branch (tree.rs:85-107) */
                next_tok(&c, &tb, &tl);
                if (parse_child(tb, tl, &nd->no_is_node, &nd->no))
                    return -1;
                next_tok(&c, &tb, &tl);
                if (parse_child(tb, tl, &nd->yes_is_node, &nd->yes))
                    return -1;
            }
        } else {
            return -1;
        }
    }
    return 0;
}

static int node_index_by_id(const jbo_tree *t, long id)
{
    for (int i = 0; i < t->nnode; i++)
        if (t->nodes[i].id == id)
            return i;
    return -1;
}

/* Tree::search_node (src/model/voice/tree.rs:14-27): start at node 0. */
static int tree_search(const jbo_model *m, const jbo_tree *t, const char *label)
{
    if (t->nnode == 0)
        return t->single_leaf;
    int ni = 0;
    for (;;) {
        const jbo_node *nd = &t->nodes[ni];
        int yes = question_test(&m->qs[nd->q], label);
        int isn = yes ? nd->yes_is_node : nd->no_is_node;
        long v = yes ? nd->yes : nd->no;
        if (!isn)
            return (int)v;
        ni = node_index_by_id(t, v);
        if (ni < 0)
            return -1;
    }
}

static int parse_model(jbo_model *m, const uint8_t *data, size_t ndata, long t0, long t1,
                       long p0, long p1, int pdf_len)
{
    if (t1 >= (long)ndata || p1 >= (long)ndata)
        return -1;
    if (parse_tree_text(m, (const char *)data + t0, (const char *)data + t1 + 1))
        return -1;
    m->pdf_len = pdf_len;
    m->npdf = (int *)calloc((size_t)m->ntree, sizeof(int));
    m->pdf = (float **)calloc((size_t)m->ntree, sizeof(float *));
    const uint8_t *p = data + p0;
    for (int k = 0; k < m->ntree; k++) {
        uint32_t n;
        memcpy(&n, p, 4); /* LE host assumed */
        p += 4;
        m->npdf[k] = (int)n;
    }
    for (int k = 0; k < m->ntree; k++) {
        size_t cnt = (size_t)m->npdf[k] * (size_t)pdf_len;
        m->pdf[k] = (float *)malloc(cnt * sizeof(float));
        memcpy(m->pdf[k], p, cnt * 4);
        p += cnt * 4;
    }
    if (p != data + p1 + 1)
        return -1;
    return 0;
}

static void split_csv(const char *s, char out[][32], int *n, int cap)
{
    *n = 0;
    while (*s && *n < cap) {
        const char *e = strchr(s, ',');
        size_t l = e ? (size_t)(e - s) : strlen(s);
        if (l > 31)
            l = 31;
        memcpy(out[*n], s, l);
        out[*n][l] = 0;
        (*n)++;
        if (!e)
            break;
        s = e + 1;
    }
}

jbo_voice *jbo_voice_load_bytes(const uint8_t *bytes, size_t n)
{
    const char *b = (const char *)bytes, *e = b + n;
    const char *g = NULL, *s = NULL, *p = NULL, *d = NULL;
    /* split_sections (src/model/parser/mod.rs:76-102) */
    for (const char *q = b; q + 10 < e; q++) {
        if (q != b && q[-1] != '\n')
            continue;
        if (!g && !memcmp(q, "[GLOBAL]\n", 9))
            g = q + 9;
        else if (!s && !memcmp(q, "[STREAM]\n", 9))
            s = q + 9;
        else if (!p && !memcmp(q, "[POSITION]\n", 11))
            p = q + 11;
        else if (!d && !memcmp(q, "[DATA]\n", 7)) {
            d = q + 7;
            break;
        }
    }
    if (!g || !s || !p || !d)
        return NULL;
    const char *gend = s - 9, *send = p - 11, *pend = d - 7;
    jbo_voice *v = (jbo_voice *)calloc(1, sizeof *v);
    char *t;
#define GETI(dst, key)                                                                         \
    do {                                                                                       \
        t = header_get(g, gend, key);                                                          \
        if (!t)                                                                                \
            goto fail;                                                                         \
        dst = atoi(t);                                                                         \
        free(t);                                                                               \
    } while (0)
    GETI(v->fs, "SAMPLING_FREQUENCY");
    GETI(v->fperiod, "FRAME_PERIOD");
    GETI(v->nstate, "NUM_STATES");
    GETI(v->nstream, "NUM_STREAMS");
    if (v->nstream > JBO_MAX_STREAM)
        goto fail;
    t = header_get(g, gend, "STREAM_TYPE");
    if (!t)
        goto fail;
    int nt;
    split_csv(t, v->stream_type, &nt, JBO_MAX_STREAM);
    free(t);
    /* GV_OFF_CONTEXT:"*-sil+*","*-pau+*" */
    t = header_get(g, gend, "GV_OFF_CONTEXT");
    v->gv_off.name = dupn("GV_OFF_CONTEXT", 14);
    if (t) {
        const char *c = t;
        int cap = 0;
        while (*c) {
            if (*c == '"') {
                const char *s0 = ++c;
                while (*c && *c != '"')
                    c++;
                if (v->gv_off.npat == cap) {
                    cap = cap ? cap * 2 : 4;
                    v->gv_off.pats = (char **)realloc(v->gv_off.pats, sizeof(char *) * (size_t)cap);
                }
                v->gv_off.pats[v->gv_off.npat++] = dupn(s0, (size_t)(c - s0));
            }
            if (*c)
                c++;
        }
        free(t);
    }
    const uint8_t *data = (const uint8_t *)d;
    size_t ndata = (size_t)(e - d);
    long a0, a1, b0, b1;
    char key[96];
    t = header_get(p, pend, "DURATION_PDF");
    if (!t || parse_range(t, &a0, &a1))
        goto fail;
    free(t);
    t = header_get(p, pend, "DURATION_TREE");
    if (!t || parse_range(t, &b0, &b1))
        goto fail;
    free(t);
    if (parse_model(&v->dur, data, ndata, b0, b1, a0, a1, v->nstate * 2))
        goto fail;
    v->alpha = 0.0;
    v->stage = 0;
    for (int i = 0; i < v->nstream; i++) {
        jbo_vstream *st = &v->st[i];
        const char *nm = v->stream_type[i];
#define SGETI(dst, pre)                                                                        \
    do {                                                                                       \
        snprintf(key, sizeof key, pre "[%s]", nm);                                             \
        t = header_get(s, send, key);                                                          \
        if (!t)                                                                                \
            goto fail;                                                                         \
        dst = atoi(t);                                                                         \
        free(t);                                                                               \
    } while (0)
        SGETI(st->L, "VECTOR_LENGTH");
        SGETI(st->is_msd, "IS_MSD");
        SGETI(st->W, "NUM_WINDOWS");
        SGETI(st->use_gv, "USE_GV");
        snprintf(key, sizeof key, "OPTION[%s]", nm);
        t = header_get(s, send, key);
        if (t && i == 0) {
            /* Condition::load_model reads options of stream 0 only
             * (src/engine.rs:96-119) */
            char *o = t;
            while (*o) {
                char *cm = strchr(o, ',');
                if (cm)
                    *cm = 0;
                if (!strncmp(o, "ALPHA=", 6))
                    v->alpha = strtod(o + 6, NULL);
                else if (!strncmp(o, "GAMMA=", 6))
                    v->stage = atoi(o + 6);
                else if (!strncmp(o, "LN_GAIN=", 8))
                    v->use_log_gain = atoi(o + 8);
                if (!cm)
                    break;
                o = cm + 1;
            }
        }
        free(t);
        /* windows */
        snprintf(key, sizeof key, "STREAM_WIN[%s]", nm);
        t = header_get(p, pend, key);
        if (!t)
            goto fail;
        {
            char *o = t;
            int w = 0;
            size_t nc = 0;
            while (*o && w < JBO_MAX_WIN) {
                long w0, w1;
                if (parse_range(o, &w0, &w1))
                    goto fail;
                char *txt = dupn((const char *)data + w0, (size_t)(w1 - w0 + 1));
                char *q = txt;
                long cnt = strtol(q, &q, 10);
                st->win_width[w] = (uint32_t)cnt;
                st->win_off[w] = (uint32_t)nc;
                for (long k = 0; k < cnt; k++)
                    st->win_coef[nc++] = strtod(q, &q);
                free(txt);
                w++;
                char *cm = strchr(o, ',');
                if (!cm)
                    break;
                o = cm + 1;
            }
            if (w != st->W)
                goto fail;
        }
        free(t);
        snprintf(key, sizeof key, "STREAM_PDF[%s]", nm);
        t = header_get(p, pend, key);
        if (!t || parse_range(t, &a0, &a1))
            goto fail;
        free(t);
        snprintf(key, sizeof key, "STREAM_TREE[%s]", nm);
        t = header_get(p, pend, key);
        if (!t || parse_range(t, &b0, &b1))
            goto fail;
        free(t);
        if (parse_model(&st->model, data, ndata, b0, b1, a0, a1, st->L * st->W * 2 + st->is_msd))
            goto fail;
        if (st->use_gv) {
            snprintf(key, sizeof key, "GV_PDF[%s]", nm);
            t = header_get(p, pend, key);
            if (!t || parse_range(t, &a0, &a1))
                goto fail;
            free(t);
            snprintf(key, sizeof key, "GV_TREE[%s]", nm);
            t = header_get(p, pend, key);
            if (!t || parse_range(t, &b0, &b1))
                goto fail;
            free(t);
            if (parse_model(&st->gv, data, ndata, b0, b1, a0, a1, st->L * 2))
                goto fail;
        }
    }
    return v;
fail:
    jbo_voice_free(v);
    return NULL;
}

jbo_voice *jbo_voice_load(const char *path)
{
    FILE *f = fopen(path, "rb");
    if (!f)
        return NULL;
    fseek(f, 0, SEEK_END);
    long n = ftell(f);
    fseek(f, 0, SEEK_SET);
    uint8_t *b = (uint8_t *)malloc((size_t)n);
    if (fread(b, 1, (size_t)n, f) != (size_t)n) {
        fclose(f);
        free(b);
        return NULL;
    }
    fclose(f);
    jbo_voice *v = jbo_voice_load_bytes(b, (size_t)n);
    free(b);
    return v;
}

static void free_model(jbo_model *m)
{
    for (int i = 0; i < m->nq; i++) {
        free(m->qs[i].name);
        for (int k = 0; k < m->qs[i].npat; k++)
            free(m->qs[i].pats[k]);
        free(m->qs[i].pats);
    }
    free(m->qs);
    for (int i = 0; i < m->ntree; i++) {
        free(m->trees[i].nodes);
        if (m->pdf)
            free(m->pdf[i]);
    }
    free(m->trees);
    free(m->npdf);
    free(m->pdf);
}

void jbo_voice_free(jbo_voice *v)
{
    if (!v)
        return;
    free_model(&v->dur);
    for (int i = 0; i < JBO_MAX_STREAM; i++) {
        free_model(&v->st[i].model);
        free_model(&v->st[i].gv);
    }
    free(v->gv_off.name);
    for (int k = 0; k < v->gv_off.npat; k++)
        free(v->gv_off.pats[k]);
    free(v->gv_off.pats);
    free(v);
}

/* ---- accessors ---------------------------------------------------------- */
const jbo_model *jbo_model_of(const jbo_voice *v, int kind)
{
    if (kind == 0)
        return &v->dur;
    if (kind >= 1 && kind <= 3)
        return &v->st[kind - 1].model;
    if (kind >= 4 && kind <= 6)
        return &v->st[kind - 4].gv;
    return NULL;
}
int jbo_voice_sampling_frequency(const jbo_voice *v) { return v->fs; }
int jbo_voice_fperiod(const jbo_voice *v) { return v->fperiod; }
int jbo_voice_nstate(const jbo_voice *v) { return v->nstate; }
int jbo_voice_nstream(const jbo_voice *v) { return v->nstream; }
double jbo_voice_alpha(const jbo_voice *v) { return v->alpha; }
int jbo_voice_stage(const jbo_voice *v) { return v->stage; }
int jbo_voice_vector_length(const jbo_voice *v, int s) { return v->st[s].L; }
int jbo_voice_num_windows(const jbo_voice *v, int s) { return v->st[s].W; }
int jbo_voice_is_msd(const jbo_voice *v, int s) { return v->st[s].is_msd; }
int jbo_voice_use_gv(const jbo_voice *v, int s) { return v->st[s].use_gv; }
int jbo_voice_window(const jbo_voice *v, int s, int w, double *coef, int cap)
{
    const jbo_vstream *st = &v->st[s];
    int n = (int)st->win_width[w];
    for (int i = 0; i < n && i < cap; i++)
        coef[i] = st->win_coef[st->win_off[w] + (uint32_t)i];
    return n;
}
int jbo_voice_ntree(const jbo_voice *v, int kind) { return jbo_model_of(v, kind)->ntree; }
int jbo_voice_npdf(const jbo_voice *v, int kind, int tree)
{
    return jbo_model_of(v, kind)->npdf[tree];
}
int jbo_voice_pdf_len(const jbo_voice *v, int kind) { return jbo_model_of(v, kind)->pdf_len; }
const float *jbo_voice_pdf(const jbo_voice *v, int kind, int tree, int idx1)
{
    const jbo_model *m = jbo_model_of(v, kind);
    return m->pdf[tree] + (size_t)(idx1 - 1) * (size_t)m->pdf_len;
}

/* Model::get_index (src/model/voice/model.rs:51-74) */
int jbo_model_get_index(const jbo_model *m, int state_index, const char *label, int *tree_pos,
                        int *pdf_index)
{
    int ti = -1;
    for (int i = 0; i < m->ntree; i++)
        if (m->trees[i].state == state_index) {
            ti = i;
            break;
        }
    const jbo_tree *t = &m->trees[ti < 0 ? 0 : ti];
    *tree_pos = ti;
    *pdf_index = tree_search(m, t, label);
    return (*pdf_index > 0) ? 0 : -1;
}

int jbo_voice_get_index(const jbo_voice *v, int kind, int state_index, const char *label,
                        int *tree_state, int *pdf_index)
{
    int tp;
    int r = jbo_model_get_index(jbo_model_of(v, kind), state_index, label, &tp, pdf_index);
    *tree_state = tp < 0 ? -1 : tp + 2;
    return r;
}

int jbo_gv_off(const jbo_voice *v, const char *label) { return question_test(&v->gv_off, label); }
