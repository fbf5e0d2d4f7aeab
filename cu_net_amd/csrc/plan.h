// Host-side execution plan of the Coupled U-Net: state layout (reference state_dict order),
// tensor table, node list and workspace layout.  Pure C++ (no device calls) so it can be
// built, described and checked on a machine without a GPU.
#pragma once
#include <cstdint>
#include <deque>
#include <map>
#include <string>
#include <vector>

#include "../../include/cunet.h"

namespace cunet {

struct StateEntry {
    std::string name;
    int kind;                 // 0 param, 1 buffer, 2 counter
    std::vector<int64_t> shape;
    int64_t offset;           // element offset inside its arena
    int64_t numel;
    int bucket = 0;           // gradient bucket (U-Net index; layer_num = stem)
};

struct TensorInfo {
    std::string name;
    int N, H, W, C, ld;
    int64_t rows() const { return (int64_t)N * H * W; }
    int64_t act = -1;         // float offset in workspace
    int64_t grad = -1;        // float offset in workspace (training)
    int64_t stats = -1;       // double offset in the zeroed region ([2][C]) or -1
    int gld16 = 0;            // leading dimension of the gradient tensor when gradient tensors are stored as bf16 (heads: K padded to a 32-multiple)
    int cfirst = 0, ccount = 0;   // slice of Plan::contribs: the conv nodes that read this tensor (backward gather)
};

struct Contrib { int node, seg; };   // node's segs[seg] is the tensor: its dz slice contributes to the tensor's gradient

struct BnInfo {
    std::string name;         // module path, e.g. hg.down_blocks.0.layers.0.norm1
    int C;
    int64_t gamma, beta;      // param arena
    int64_t rmean, rvar;      // buffer arena
    int64_t counter;          // counter arena
    bool ckpt;                // re-run by torch.utils.checkpoint in the reference's backward
};

struct ConvInfo {
    std::string name;         // module path of the conv
    int Cout, Cin, taps;
    int64_t w;                // param arena
    int64_t wF = -1, wB = -1; // workspace float offsets of the repacked operands
    int64_t tern = -1;        // uint64 offset of the AND-popcount bit masks (wpos, then wneg) in the mask region; -1: not a QuanInput site
    int KpadF, NpadF, KpadB, NpadB;
};

struct SegRef {
    int tensor;
    int ups;
};

enum NodeType { N_STEM_CONV = 0, N_STEM_BNPOOL = 1, N_CONV = 2, N_POOL = 3 };

struct Node {
    NodeType type;
    std::string name;
    int bn = -1;              // index into bns
    int conv = -1;            // index into convs
    std::vector<SegRef> segs; // inputs (N_CONV); segs[0] is the single input for pool / stem bn-pool
    int out = -1;             // output tensor
    int taps = 1;
    int head = -1;            // >= 0: this conv is heat-map head number `head` (output index)
    int Ccat = 0;             // channels of the concat
    int64_t red = -1;         // double offset in the zeroed region: [2][Ccat] backward reductions
    int bucket = 0;           // gradient bucket this node's parameters live in
    int64_t dz = -1;          // float offset of this node's own [M][Ccat] dz buffer (training)
    int pair = 0;             // 1: this node (the ahead adapter of a down block) and the NEXT node (the skip adapter over the same
                              // concat, models/cu_net.py:139-142) may run as one launch, forward and data gradient
    // LDS-staged 1x1 weight gradient (wgrad3): pixel splits, rows per split, float offset of the [S][Cout][Ccat]
    // partial tiles inside the per-bucket partial region, index into the reduce table; wg3_S == 0: not eligible
    int wg3_S = 0, wg3_rows = 0, wg3_entry = -1;
    int wg3_cap = 0;                      // partial tiles the node's slice of the partial region holds (>= wg3_S; the fused data + weight gradient of a
                                          // 1x1 node writes one per ROW BLOCK of its launch, a count the runtime fixes at bind)
    int fuse_ok = 0;                      // 1: fp32 1x1 node whose weight gradient may be computed by its data-gradient launch (planner option fuse_wgrad)
    int wg3_S16 = 0, wg3_rows16 = 0;
    int wg3_wpi = 0;                      // stem: workgroups per image (wg3_S = N * wg3_wpi, wg3_rows = output rows per workgroup)      // the same with bf16 gradient tensors: those kernels are HBM-bound, fewer and longer splits (less partial traffic) win
    int64_t wg3_part = -1;
};

struct PlannerOptions { int wgrad3_min_rows = 0, wgrad3_min_chunks = 2, wgrad3_max_splits = 0, wgrad3_min_chunks_bf16 = 4, wgrad3_max_splits_bf16 = 128, wgrad3_stem = 1, conv3x3_ring_min_rows = 512, wgrad_fork_group = 0, wgrad_fork_group_bf16 = 8, fwd_fork_min_w = 0, pair_adapters = 1, heads_on_side = 2, dgrad_nt = 4, wgrad_bf16_dma = 1, fuse_wgrad = 0, dgrad_prefetch = 1, dgrad_rows = -1, f32_split = 1, dgrad3_nt = 1, dgrad3_ring = 768, stem_split = 1, dgrad_rows_v = 2, popcount_pixels = 1, stem_fuse_dz = 1, stem_wgrad_split = 1, fuse_pool_gather = 1, fuse_z_gather = 0, stem_wgrad_caller = 0, wgrad_split_planes = 0, stem_wgrad_planes = 1; };
PlannerOptions& planner_options();

struct Plan {
    cunet_cfg cfg;
    PlannerOptions opts;      // snapshot of the process-wide options taken when the plan was created (cunet_set_planner_option later
                              // does not change kernel selection of a live plan)
    // elements of one wgrad3 partial tile == of the node's weight tensor
    int64_t wg3_region = 0;               // floats of ONE partial region; there are two, used alternately by the buckets in backward's order
    int bucket_position(int b) const { return b == cfg.layer_num ? cfg.layer_num : cfg.layer_num - 1 - b; }      // backward visits U-Nets L-1 .. 0, then the stem
    int64_t wg3_numel(const Node& n) const { return n.type == N_STEM_CONV ? (int64_t)convs[n.conv].Cout * convs[n.conv].Cin : (int64_t)convs[n.conv].Cout * n.Ccat * n.taps; }
    std::vector<int> anchors;
    std::vector<StateEntry> state;
    std::map<std::string, int> state_index;
    int64_t n_params = 0, n_buffers = 0, n_counters = 0;
    std::vector<int64_t> bucket_begin, bucket_count;   // float ranges of the parameter arena

    std::vector<TensorInfo> tensors;
    std::vector<BnInfo> bns;
    std::vector<ConvInfo> convs;
    std::vector<Node> nodes;
    std::vector<Contrib> contribs;
    std::vector<int> head_tensors;   // per output index: tensor id of the NHWC heat map

    // workspace layout (bytes)
    int64_t off_repack_tab = 0, off_runstat_tab = 0;
    int64_t off_zero = 0, zero_bytes = 0;     // fp64 region cleared at the start of every forward
    int64_t n_zero_doubles = 0;
    int64_t loss_acc = -1;                    // double offset inside the zero region
    int64_t off_floats = 0;                   // base of the float region (byte offset)
    int64_t n_floats_infer = 0, n_floats_train = 0;
    int64_t dz_off = -1, target_off = -1;     // float offsets (training)
    int64_t ws_bytes_infer = 0, ws_bytes_train = 0;
    int64_t off_bf16 = 0, ws_bytes_bf16 = 0;     // bf16 inference: a bf16 arena of n_floats_infer elements behind the fp32 inference layout
    int64_t off_bf16_train = 0, ws_bytes_bf16_train = 0;   // the same arena behind the training layout (bf16 activations, fp32 gradients)
    int n_runstat = 0;
    int64_t off_planes = 0;                      // bit-plane records of the AND-popcount forward: (largest site rows + 1) x 128 bytes, reused site after site
    int64_t off_ternpack_tab = 0, off_tern = 0;   // pack table / bit-mask region of the quantised-input mode (bytes)
    int n_tern_sites = 0;
    int64_t off_wgred_tab = 0;                // reduce table of the wgrad3 nodes (bytes), entries grouped by bucket
    int n_wgred = 0;
    std::vector<int> wgred_first, wgred_count;   // per bucket: slice of the reduce table
    std::vector<int> wgred_maxnumel;             // per bucket: largest numel in its slice

    std::string json;
    std::string error;

    bool build(const cunet_cfg& c);
    int tensor_by_name(const std::string& n) const;

   private:
    int64_t param(const std::string& n) const;
    int add_tensor(const std::string& name, int N, int H, int W, int C, bool stats);
    int add_bn(const std::string& path, bool ckpt);
    int add_conv(const std::string& path, int taps, bool need_bwd);
    int conv_node(const std::string& name, const std::string& bn_path, const std::string& conv_path,
                  const std::vector<SegRef>& segs, int taps, bool ckpt, int H, int W, int head, bool out_stats);
    void build_state();
    void block_entries(const std::string& prefix, int in_num, bool requires_skip, bool is_up);
    void bn_entries(const std::string& prefix, int c, int bucket);
    void add_state(const std::string& name, int kind, std::vector<int64_t> shape, int bucket);
    void assign_param_offsets();
    void layout_workspace();
    void describe();
};

}  // namespace cunet
