// jb_voice.h -- host-side voice model (cold path): .htsvoice container, decision
// trees, pdf tables.  Mirrors the *data model* of /root/reference/src/model/voice
// (Voice{metadata, duration_model, stream_models}, Model{trees, pdf}) with trees
// flattened to index arrays at load time.
#pragma once
#include <cstdint>
#include <memory>
#include <optional>
#include <stdexcept>
#include <string>
#include <string_view>
#include <vector>

namespace jb {

struct ModelError : std::runtime_error {
    using std::runtime_error::runtime_error;
};

// HTS question: OR of glob patterns ('*', '?') over the full-context label string.
// Nearly every pattern of a voice is "*X*", "X*" or "*X" with a literal X: those are classified
// once (compile()) and tested with find / starts_with / ends_with; the rest go through the
// general matcher.
struct Question {
    enum Kind : uint8_t { Glob, Contains, Prefix, Suffix, Exact, Any };
    std::vector<std::string> patterns;
    std::vector<std::pair<Kind, std::string>> compiled; // same order as patterns
    void compile();
    bool test(std::string_view label) const;
};

struct Model;
// Question results of ONE label, per model: the five state trees of a stream ask largely the
// same questions, so each is evaluated at most once per (model, label).
struct QuestionMemo {
    struct Slot {
        const Model *m = nullptr;
        std::vector<int8_t> v; // -1 unknown, 0 / 1
    };
    std::vector<Slot> slots;
    std::vector<int8_t> *of(const Model *m, size_t nq);
    void reset(); // new label
};

bool glob_match(std::string_view pat, std::string_view s);

struct TreeNode {
    int32_t question; // index into Model::questions
    int32_t yes, no;  // >= 0: node index; < 0: leaf, pdf index = -value (1-based)
};

struct Tree {
    int state = 0;
    std::vector<TreeNode> nodes; // empty => single leaf
    int single_leaf = 0;
    // 1-based pdf index; memo (optional) caches question results of this label
    int search(const std::vector<Question> &qs, std::string_view label, std::vector<int8_t> *memo = nullptr) const;
};

// Model (src/model/voice/model.rs:12-82)
struct Model {
    std::vector<Question> questions;
    std::vector<Tree> trees;
    int pdf_len = 0;
    std::vector<int> npdf;                // per tree
    std::vector<std::vector<float>> pdf;  // per tree: npdf*pdf_len (LE f32 as stored)

    // get_index: (tree position or -1 when no tree has that state, 1-based pdf index)
    void get_index(int state_index, std::string_view label, int &tree_pos, int &pdf_index,
                   QuestionMemo *memo = nullptr) const;
    // get_parameter: pointer to pdf_len floats (first half means, second half variances,
    // optional trailing MSD weight: ModelParameter::from_linear, model.rs:99-109)
    const float *get_parameter(int state_index, std::string_view label, QuestionMemo *memo = nullptr) const;
};

struct StreamModel {
    std::string name;
    int vector_length = 0, num_windows = 0;
    bool is_msd = false, use_gv = false;
    std::vector<std::string> options;
    std::vector<std::vector<double>> windows;
    Model stream;
    std::optional<Model> gv;
};

struct GlobalMeta {
    std::string hts_voice_version;
    int sampling_frequency = 0, frame_period = 0, num_states = 0, num_streams = 0;
    std::vector<std::string> stream_type;
    std::string fullcontext_format, fullcontext_version;
    std::vector<std::string> gv_off_patterns;
    bool operator==(const GlobalMeta &o) const;
};

struct Voice {
    GlobalMeta meta;
    Question gv_off;
    Model duration;
    std::vector<StreamModel> streams;
};

// parse_htsvoice (src/model/parser/mod.rs:58-74); throws ModelError.
std::shared_ptr<Voice> parse_htsvoice(const uint8_t *bytes, size_t n);
std::shared_ptr<Voice> load_htsvoice(const std::string &path);

} // namespace jb
