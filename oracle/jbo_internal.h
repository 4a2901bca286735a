/* oracle/jbo_internal.h -- CPU ORACLE internals (test infrastructure, NOT the product). */
#ifndef JBO_INTERNAL_H
#define JBO_INTERNAL_H
#include "jbo.h"

#define JBO_MAX_WIN 8

typedef struct {
    char *name;
    int npat;
    char **pats;
} jbo_question;

typedef struct {
    long id;
    int q;
    int yes_is_node, no_is_node;
    long yes, no; /* node id, or 1-based pdf index */
} jbo_node;

typedef struct {
    int state;
    int nnode;
    jbo_node *nodes;
    int single_leaf; /* pdf index when nnode == 0 */
} jbo_tree;

typedef struct {
    int nq;
    jbo_question *qs;
    int ntree;
    jbo_tree *trees;
    int pdf_len;
    int *npdf;
    float **pdf; /* [tree][npdf*pdf_len], LE f32 as stored (widened on use, parser/model/mod.rs:49) */
} jbo_model;

typedef struct {
    int L, W, is_msd, use_gv;
    uint32_t win_width[JBO_MAX_WIN];
    uint32_t win_off[JBO_MAX_WIN];
    double win_coef[64];
    jbo_model model;
    jbo_model gv;
} jbo_vstream;

struct jbo_voice {
    int fs, fperiod, nstate, nstream;
    char stream_type[JBO_MAX_STREAM][32];
    double alpha;
    int stage;
    int use_log_gain;
    jbo_question gv_off;
    jbo_model dur;
    jbo_vstream st[JBO_MAX_STREAM];
};

int jbo_glob(const char *pat, const char *s);
const jbo_model *jbo_model_of(const jbo_voice *v, int kind);
int jbo_model_get_index(const jbo_model *m, int state_index, const char *label, int *tree_pos,
                        int *pdf_index);

#endif
