// The RawFormer handle (internal): parameter registry, packed-weight plan.  Shared by the forward schedule (rf_model.hip) and
// the training schedule (rf_trainstep.hip).
#pragma once
#include <string>
#include <unordered_map>
#include <vector>
#include "rf_common.h"

struct Param {
    std::string name;
    int64_t shape[4];
    int ndim;
    const float* ptr;
    size_t numel() const {
        size_t n = 1;
        for (int i = 0; i < ndim; ++i) n *= (size_t)shape[i];
        return n;
    }
};

enum PackKind { PK_1x1, PK_3x3, PK_CONVT, PK_1x1_B3 };
struct PackItem {
    int param;       // index of the raw weight
    PackKind kind;
    size_t offset;   // floats into the packed buffer
    size_t floats;
};

struct rf_handle {
    rf_config cfg;
    std::vector<Param> params;
    std::unordered_map<std::string, int> index;
    std::vector<PackItem> packs;
    std::unordered_map<std::string, int> pack_index;   // weight name -> packs[]
    std::unordered_map<std::string, int> pack3_index;  // weight name -> packs[] entry of its b3 form
    size_t packed_floats = 0;
    size_t upcat_offset[3] = {0, 0, 0};   // composed decoder-step weights (rf_upcat.hip), floats into the packed buffer
    // composed stage tails (rf_flca.hip pack_tail): per Conv_Transformer stage 1..7, floats into the packed buffer of [Wb W2 | b']
    // (0 = stage not composed) and, plain variant only, of the static b3 weights [Wa | Wb | Wb W2]
    size_t tail_offset[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tail3_offset[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const float* packed = nullptr;   // caller memory, valid after rf_pack_params
    std::vector<size_t> flat_offset;  // float offset of every parameter in the flat parameter / gradient buffers (training)
    size_t flat_floats = 0;
    // spatial shard of one frame (rf_set_shard): interior rows [y_lo, y_hi) of the local window and the frame's total rows, in
    // packed (level-0) rows; `allreduce` sums a float buffer over the ranks on the given stream
    int shard_y_lo = 0, shard_y_hi = 0, shard_total_rows = 0;
    void (*shard_allreduce)(void* user, float* buf, size_t n, int op, void* stream) = nullptr;
    void* shard_user = nullptr;
    // branch stream (rf_forward): the guidance pyramid and each stage's FLCA / 3x3 branch run beside the TransformerBlock;
    // forked from and joined into the caller's stream with the two events, created on first use
    hipStream_t side = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    bool side_failed = false;
    // training: notification that a range of the flat gradient buffer is final (rf_set_grad_ready)
    void (*grad_ready)(void* user, size_t offset, size_t count, void* stream) = nullptr;
    void* grad_ready_user = nullptr;
};


inline const float* rf_param_ptr(const rf_handle* h, const std::string& name) { return h->params[h->index.at(name)].ptr; }
inline int rf_param_index(const rf_handle* h, const std::string& name) { return h->index.at(name); }
