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
struct Question {
    std::vector<std::string> patterns;
    bool test(std::string_view label) const;
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
    int search(const std::vector<Question> &qs, std::string_view label) const; // 1-based pdf index
};

// Model (src/model/voice/model.rs:12-82)
struct Model {
    std::vector<Question> questions;
    std::vector<Tree> trees;
    int pdf_len = 0;
    std::vector<int> npdf;                // per tree
    std::vector<std::vector<float>> pdf;  // per tree: npdf*pdf_len (LE f32 as stored)

    // get_index: (tree position or -1 when no tree has that state, 1-based pdf index)
    void get_index(int state_index, std::string_view label, int &tree_pos, int &pdf_index) const;
    // get_parameter: pointer to pdf_len floats (first half means, second half variances,
    // optional trailing MSD weight: ModelParameter::from_linear, model.rs:99-109)
    const float *get_parameter(int state_index, std::string_view label) const;
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
