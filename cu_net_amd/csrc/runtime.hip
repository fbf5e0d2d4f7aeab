// C-ABI entry points (include/cunet.h) and the executor that walks a Plan and enqueues the HIP
// kernels on the caller's stream.  Owns no device memory.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/cunet.h"
#include "kernels.h"
#include "plan.h"

enum { PC_C1F = 0, PC_C3F, PC_STEMF, PC_C1D, PC_C3D, PC_C1W, PC_C3W, PC_STEMW, PC_APPLY, PC_POOLF, PC_POOLB,
       PC_STEMBPF, PC_STEMBPB, PC_MISC, PC_C1F16, PC_C3F16, PC_C1D16, PC_C3D16, PC_C1W16, PC_TERN, PC_C3W16, CUNET_PROF_NCLS };

using namespace cunet;

struct cunet_plan {
    Plan plan;
    // bound, caller-owned device memory
    float* params = nullptr;
    float* grads = nullptr;
    float* buffers = nullptr;
    int64_t* counters = nullptr;
    char* ws = nullptr;
    int64_t ws_bytes = 0;
    int bound_training = 0;
    int num_cus = 256;
    // host copies of the device tables (kept alive for async uploads)
    std::vector<RepackEntry> repack;
    std::vector<RunStatEntry> runstat;
    std::vector<WgReduceEntry> wgred;
    // quantised-input mode (cunet_set_quant_input): per node the QuanInput bit width (0 = none) and whether its forward
    // runs on the AND-popcount kernel; the pack table of those nodes' convs
    int qin_bits = 0;
    std::vector<int> node_qin, node_tern;
    std::vector<TernPackEntry> ternpack;
    int ternpack_dirty = 0;
    float* fused_loss_out = nullptr;  // cunet_loss_mse_fused: the next training forward computes the loss in its head epilogues
    int tern_live = 0;               // cunet_set_popcount_live: the caller vouches that those convs' weights ARE ternary right now
    int fz_node = -1;                // backward: the node whose data gradient is being launched with its output gradient's gather folded into the operand load (planner option fuse_z_gather)
    bool stem_fused_now = false;     // this backward pass: the stem's weight gradient computes d(loss)/d(conv0 output) itself (planner option stem_fuse_dz)
    // call-order state
    int fwd_training_done = 0;
    int loss_done = 0;
    const float* last_x = nullptr;   // image of the last training forward (needed by the stem weight gradient)
    // internal side stream: weight gradients run concurrently with the data-gradient chain
    hipStream_t side = nullptr;
    std::vector<hipEvent_t> fork_ev;     // one per node: "d(loss)/d(out) of node k is ready"
    std::vector<hipEvent_t> done_ev;     // forward: node k (run on the side stream) has finished
    std::vector<int> pending;            // forward: tensor -> node whose side-stream result it is, or -1
    hipEvent_t join_ev = nullptr;
    int use_side = 1;
    // fp32 1x1 nodes whose weight gradient is computed by their data-gradient launch (conv_body's fused loop; planner option fuse_wgrad):
    // per node the number of partial tiles that launch writes (0 = the node keeps its own weight-gradient launch), fixed at bind
    std::vector<int> fused_S;
    hipEvent_t red_ev[2] = {nullptr, nullptr};      // "the reduce of the bucket at backward position p has run", by parity of p (two partial regions)
    hipEvent_t bucket_ev = nullptr;                 // "every data gradient of the bucket is enqueued" (its fused partial tiles are complete behind it)
    // optional per-kernel-class timing with HIP events on the launch stream (bench.py roofline)
    int prof_mode = 0;               // 0 off, 1 every class, 2 only prof_cls
    int prof_cls = -1;
    struct ProfRec { hipEvent_t a, b; int cls, on_side; double flops, bytes; };
    std::vector<ProfRec> prof_pending;
    std::vector<hipEvent_t> prof_pool;
    // [0]: launches on the caller's stream, [1]: launches on the internal low-priority side stream (their durations include the
    // time they are switched out for the caller's kernels)
    double prof_ms[2][CUNET_PROF_NCLS] = {{0}};
    double prof_flops[2][CUNET_PROF_NCLS] = {{0}};
    double prof_bytes[2][CUNET_PROF_NCLS] = {{0}};
    long prof_count[2][CUNET_PROF_NCLS] = {{0}};
};

// Does node `n` take the LDS-staged partial-tile weight gradient in gradient-storage mode `xmode`?  (One predicate for the
// dispatch in bwd_node and for the reduce tables built at bind.)
static bool wg3_active(const Plan& P, const Node& n, int xmode) {
    if (n.wg3_S <= 0) return false;
    if (n.type == N_STEM_CONV) return true;                      // (fp32 image and fp32 dY in every mode)
    // 3x3 ring kernel (bf16 x / dY are widened to fp32 on the way into LDS).  Alone on the GPU it takes 47 / 29 / 21 us at
    // W = 32 / 16 / 8 against 31 / 16 / 14 us of the per-wave kernel (two barriers per image row of W/2 MFMA steps) and wins
    // at W = 64 (88 vs 106 us); next to the fp32 data-gradient chain it is ahead at every width (fewer atomics, one read
    // of X: 3155 vs 3083 img/s), next to the much shorter bf16 chain only where it is also faster alone.
    // With bf16 x AND dY the contraction runs on bf16 MFMA (wgrad3_3x3_bf16_kernel, W = 16 / 32 / 64).
    if (n.taps == 9) {
        const int W = P.tensors[n.out].W;
        return xmode != 2 || W == 64 || W == 32 || W == 16;
    }
    if (xmode == 2) {                                            // bf16 MFMA variant: 16-byte pieces of 8 bf16 channels
        if (n.wg3_rows16 % 64) return false;
        for (auto& sr : n.segs)
            if (P.tensors[sr.tensor].C % 8 || P.tensors[sr.tensor].ld % 8) return false;
    }
    return true;
}

static const char* kProfNames[CUNET_PROF_NCLS] = {
    "conv1x1_fwd", "conv3x3_fwd", "stem_conv_fwd", "conv1x1_bwd_data", "conv3x3_bwd_data", "conv1x1_bwd_weight",
    "conv3x3_bwd_weight", "stem_bwd_weight", "bn_bwd_apply", "pool_fwd", "pool_bwd", "stem_bnpool_fwd",
    "stem_bnpool_bwd", "misc",
    // the same node classes when they run on bf16 MFMA (bf16 storage modes): priced against the bf16 peak
    "conv1x1_fwd_bf16", "conv3x3_fwd_bf16", "conv1x1_bwd_data_bf16", "conv3x3_bwd_data_bf16", "conv1x1_bwd_weight_bf16",
    "conv_fwd_popcount", "conv3x3_bwd_weight_bf16"};

static hipError_t prof_begin(cunet_plan* h, int cls, hipStream_t s, int& slot) {
    slot = -1;
    if (h->prof_mode == 0 || (h->prof_mode == 2 && cls != h->prof_cls)) return hipSuccess;
    cunet_plan::ProfRec r{};
    for (hipEvent_t* e : {&r.a, &r.b}) {
        if (!h->prof_pool.empty()) { *e = h->prof_pool.back(); h->prof_pool.pop_back(); }
        else { hipError_t er = hipEventCreate(e); if (er != hipSuccess) return er; }
    }
    r.cls = cls;
    r.on_side = (h->side != nullptr && s == h->side) ? 1 : 0;
    h->prof_pending.push_back(r);
    slot = (int)h->prof_pending.size() - 1;
    return hipEventRecord(r.a, s);
}
static hipError_t prof_end(cunet_plan* h, int slot, double flops, double bytes, hipStream_t s) {
    if (slot < 0) return hipSuccess;
    h->prof_pending[slot].flops = flops;
    h->prof_pending[slot].bytes = bytes;
    return hipEventRecord(h->prof_pending[slot].b, s);
}
// the launch behind prof_begin did not happen: withdraw its record (the newest one only)
static void prof_cancel(cunet_plan* h, int slot) {
    if (slot < 0 || slot != (int)h->prof_pending.size() - 1) return;
    h->prof_pool.push_back(h->prof_pending[slot].a);
    h->prof_pool.push_back(h->prof_pending[slot].b);
    h->prof_pending.pop_back();
}
#define PROF_ON(st, cls, flops, bytes, expr)               \
    do {                                                   \
        int slot_;                                         \
        HIPCHK(prof_begin(h, (cls), (st), slot_));         \
        HIPCHK(expr);                                      \
        HIPCHK(prof_end(h, slot_, (flops), (bytes), (st)));\
    } while (0)
#define PROF(cls, flops, bytes, expr) PROF_ON(s, cls, flops, bytes, expr)


static thread_local std::string g_err;
static int fail(int code, const std::string& msg) { g_err = msg; return code; }

#define HIPCHK(expr)                                                                           \
    do {                                                                                       \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess)                                                                  \
            return fail(CUNET_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));     \
    } while (0)

static int plan_fused_wgrads(cunet_plan* h);      // (defined behind the executor helpers)

extern "C" {

const char* cunet_last_error(void) { return g_err.c_str(); }
const char* cunet_version(void) { return "cunet-hip 0.2 (gfx950; fp32 mode: split-bf16 or fp32 MFMA by planner option f32_split; bf16 storage: bf16 MFMA)"; }

// name -> member of PlannerOptions (one table for the setter and the getter)
static int* planner_option_slot(PlannerOptions& o, const std::string& n) {
    if (n == "wgrad3_min_rows") return &o.wgrad3_min_rows;
    if (n == "wgrad3_min_chunks") return &o.wgrad3_min_chunks;
    if (n == "wgrad3_max_splits") return &o.wgrad3_max_splits;
    if (n == "wgrad3_min_chunks_bf16") return &o.wgrad3_min_chunks_bf16;
    if (n == "wgrad3_max_splits_bf16") return &o.wgrad3_max_splits_bf16;
    if (n == "wgrad3_stem") return &o.wgrad3_stem;
    if (n == "conv3x3_ring_min_rows") return &o.conv3x3_ring_min_rows;
    if (n == "wgrad_fork_group") return &o.wgrad_fork_group;
    if (n == "wgrad_fork_group_bf16") return &o.wgrad_fork_group_bf16;
    if (n == "fwd_fork_min_w") return &o.fwd_fork_min_w;
    if (n == "pair_adapters") return &o.pair_adapters;
    if (n == "heads_on_side") return &o.heads_on_side;
    if (n == "dgrad_nt") return &o.dgrad_nt;
    if (n == "wgrad_bf16_dma") return &o.wgrad_bf16_dma;
    if (n == "fuse_wgrad") return &o.fuse_wgrad;
    if (n == "dgrad_prefetch") return &o.dgrad_prefetch;
    if (n == "dgrad_rows") return &o.dgrad_rows;
    if (n == "dgrad_rows_v") return &o.dgrad_rows_v;
    if (n == "popcount_pixels") return &o.popcount_pixels;
    if (n == "stem_fuse_dz") return &o.stem_fuse_dz;
    if (n == "stem_wgrad_split") return &o.stem_wgrad_split;
    if (n == "f32_split") return &o.f32_split;
    if (n == "dgrad3_nt") return &o.dgrad3_nt;
    if (n == "dgrad3_ring") return &o.dgrad3_ring;
    if (n == "stem_split") return &o.stem_split;
    if (n == "fuse_pool_gather") return &o.fuse_pool_gather;
    if (n == "fuse_z_gather") return &o.fuse_z_gather;
    if (n == "stem_wgrad_caller") return &o.stem_wgrad_caller;
    if (n == "wgrad_split_planes") return &o.wgrad_split_planes;
    if (n == "stem_wgrad_planes") return &o.stem_wgrad_planes;
    return nullptr;
}

int cunet_set_planner_option(const char* name, int value) {
    if (!name) return fail(CUNET_ERR_INVALID, "bad planner option");
    const std::string n(name);
    if (value < 0 && !(n == "dgrad_rows" && value == -1)) return fail(CUNET_ERR_INVALID, "bad planner option");      // (dgrad_rows = -1: its default, "by f32_split")
    int* slot = planner_option_slot(planner_options(), n);
    if (!slot) return fail(CUNET_ERR_INVALID, "unknown planner option " + n);
    *slot = value;
    return CUNET_OK;
}

int cunet_get_planner_option(const char* name, int* value) {
    if (!name || !value) return fail(CUNET_ERR_INVALID, "bad planner option");
    const int* slot = planner_option_slot(planner_options(), std::string(name));
    if (!slot) return fail(CUNET_ERR_INVALID, std::string("unknown planner option ") + name);
    *value = *slot;
    return CUNET_OK;
}

int cunet_debug_set_plan_option(cunet_plan_t* plan, const char* name, int value) {
    // only launch-time choices between bit-identical kernels may change under a live plan (everything else shaped its layout)
    if (!plan || !name || value < 0) return fail(CUNET_ERR_INVALID, "bad plan option");
    const std::string n(name);
    if (n == "wgrad_bf16_dma") plan->plan.opts.wgrad_bf16_dma = value;
    else if (n == "wgrad_split_planes") plan->plan.opts.wgrad_split_planes = value;
    else if (n == "stem_wgrad_planes") plan->plan.opts.stem_wgrad_planes = value;
    else return fail(CUNET_ERR_INVALID, std::string("not a launch-time option: ") + name);
    return CUNET_OK;
}

int cunet_plan_create(const cunet_cfg* cfg, cunet_plan_t** out) {
    if (!cfg || !out) return fail(CUNET_ERR_INVALID, "null argument");
    cunet_plan* p = new (std::nothrow) cunet_plan();
    if (!p) return fail(CUNET_ERR_NOMEM, "out of host memory");
    if (!p->plan.build(*cfg)) {
        const std::string e = p->plan.error;
        delete p;
        return fail(CUNET_ERR_INVALID, e);
    }
    *out = p;
    return CUNET_OK;
}

void cunet_plan_destroy(cunet_plan_t* plan) {
    if (!plan) return;
    for (auto e : plan->fork_ev) (void)hipEventDestroy(e);
    for (auto e : plan->done_ev) (void)hipEventDestroy(e);
    if (plan->join_ev) (void)hipEventDestroy(plan->join_ev);
    for (auto e : plan->red_ev) if (e) (void)hipEventDestroy(e);
    if (plan->bucket_ev) (void)hipEventDestroy(plan->bucket_ev);
    for (auto e : plan->prof_pool) (void)hipEventDestroy(e);
    for (auto& r : plan->prof_pending) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); }
    if (plan->side) (void)hipStreamDestroy(plan->side);
    delete plan;
}

int cunet_state_count(const cunet_plan_t* plan) { return plan ? (int)plan->plan.state.size() : 0; }

int cunet_state_entry(const cunet_plan_t* plan, int index, cunet_state_desc* out) {
    if (!plan || !out || index < 0 || index >= (int)plan->plan.state.size()) return fail(CUNET_ERR_INVALID, "bad state index");
    const StateEntry& e = plan->plan.state[index];
    std::memset(out, 0, sizeof(*out));
    std::snprintf(out->name, sizeof(out->name), "%s", e.name.c_str());
    out->kind = e.kind;
    out->ndim = (int)e.shape.size();
    for (size_t i = 0; i < e.shape.size() && i < 4; ++i) out->shape[i] = e.shape[i];
    out->offset = e.offset;
    out->numel = e.numel;
    return CUNET_OK;
}

int64_t cunet_param_numel(const cunet_plan_t* p) { return p ? p->plan.n_params : 0; }
int64_t cunet_buffer_numel(const cunet_plan_t* p) { return p ? p->plan.n_buffers : 0; }
int64_t cunet_counter_numel(const cunet_plan_t* p) { return p ? p->plan.n_counters : 0; }
int64_t cunet_workspace_bytes(const cunet_plan_t* p, int training) {
    if (!p) return 0;
    if (training == 2) return p->plan.ws_bytes_bf16;      // inference incl. the bf16 arena of cunet_forward_bf16
    if (training == 3) return p->plan.ws_bytes_bf16_train; // training incl. the bf16 activation arena
    return training ? p->plan.ws_bytes_train : p->plan.ws_bytes_infer;
}
int cunet_num_heads(const cunet_plan_t* p) { return p ? (int)p->plan.head_tensors.size() : 0; }
int cunet_loss_anchors(const cunet_plan_t* p, int32_t* anchors, int capacity) {
    if (!p || !anchors) return fail(CUNET_ERR_INVALID, "null argument");
    const int n = (int)p->plan.anchors.size();
    for (int i = 0; i < n && i < capacity; ++i) anchors[i] = p->plan.anchors[i];
    return n;
}
const char* cunet_plan_describe(const cunet_plan_t* p) { return p ? p->plan.json.c_str() : ""; }

int64_t cunet_debug_tensor_offset(const cunet_plan_t* p, const char* name, int which) {
    if (!p || !name) return -1;
    const int t = p->plan.tensor_by_name(name);
    if (t < 0) return -1;
    const TensorInfo& ti = p->plan.tensors[t];
    const int64_t f = which == 0 ? ti.act : ti.grad;
    if (f < 0) return -1;
    return p->plan.off_floats + 4 * f;
}

int cunet_bind(cunet_plan_t* h, float* params, float* grads, float* buffers, int64_t* counters, void* workspace,
               int64_t workspace_bytes, int training, void* stream) {
    if (!h || !params || !buffers || !counters || !workspace) return fail(CUNET_ERR_INVALID, "null device pointer");
    if (training && !grads) return fail(CUNET_ERR_INVALID, "training bind needs a gradient arena");
    Plan& P = h->plan;
    const int64_t need = training ? P.ws_bytes_train : P.ws_bytes_infer;
    if (workspace_bytes < need) return fail(CUNET_ERR_INVALID, "workspace too small");
    if (((uintptr_t)workspace & 255) || ((uintptr_t)params & 15) || ((uintptr_t)buffers & 15) || (grads && ((uintptr_t)grads & 15)))
        return fail(CUNET_ERR_INVALID, "device pointers must be aligned (workspace 256 B, arenas 16 B)");
    h->params = params; h->grads = grads; h->buffers = buffers; h->counters = counters;
    h->ws = (char*)workspace; h->ws_bytes = workspace_bytes; h->bound_training = training;
    int dev = 0, cus = 0;
    HIPCHK(hipGetDevice(&dev));
    HIPCHK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    h->num_cus = cus > 0 ? cus : 256;

    h->repack.clear();
    for (auto& c : P.convs) {
        RepackEntry e{};
        e.src = c.w; e.dstF = c.wF; e.dstB = c.wB;
        e.Cout = c.Cout; e.Cin = c.Cin; e.taps = c.taps;
        e.KpadF = c.KpadF; e.NpadF = c.NpadF; e.KpadB = c.KpadB; e.NpadB = c.NpadB;
        h->repack.push_back(e);
    }
    h->runstat.clear();
    for (auto& n : P.nodes) {
        if (n.type == N_CONV) {
            const BnInfo& b = P.bns[n.bn];
            int choff = 0;
            for (size_t s = 0; s < n.segs.size(); ++s) {
                const TensorInfo& t = P.tensors[n.segs[s].tensor];
                RunStatEntry e{};
                e.stats = t.stats; e.rmean = b.rmean + choff; e.rvar = b.rvar + choff;
                e.counter = s == 0 ? b.counter : -1;
                e.count = (double)t.rows(); e.C = t.C; e.times = b.ckpt ? 2 : 1;
                const double nfull = (double)t.rows() * (n.segs[s].ups ? 4.0 : 1.0);
                e.unbias = nfull > 1.0 ? nfull / (nfull - 1.0) : 1.0;
                h->runstat.push_back(e);
                choff += t.C;
            }
        } else if (n.type == N_STEM_BNPOOL) {
            const BnInfo& b = P.bns[n.bn];
            const TensorInfo& t = P.tensors[n.segs[0].tensor];
            RunStatEntry e{};
            e.stats = t.stats; e.rmean = b.rmean; e.rvar = b.rvar; e.counter = b.counter;
            e.count = (double)t.rows(); e.C = t.C; e.times = 1;
            e.unbias = e.count > 1.0 ? e.count / (e.count - 1.0) : 1.0;
            h->runstat.push_back(e);
        }
    }
    if ((int)h->runstat.size() != P.n_runstat) return fail(CUNET_ERR_STATE, "internal: running-stat table size");
    // fused data + weight gradient (fp32 gradient tensors only): which nodes, and how many partial tiles their launch writes
    h->fused_S.assign(P.nodes.size(), 0);
    if (training) {
        const int rcf = plan_fused_wgrads(h);
        if (rcf != CUNET_OK) return rcf;
    }
    // three copies of the reduce table: [0, n) fp32 activations and gradient tensors, [n, 2n) bf16 gradient tensors -- where the nodes that
    // stay on the atomic kernels in that mode (3x3 convs) have S = 0 and are skipped by the reduce --, [2n, 3n) bf16 activations with fp32
    // gradient tensors (the fp32 split counts; only copy 0 knows the fused launches' partial-tile counts)
    const int nwg = P.n_wgred > 0 ? P.n_wgred : 1;
    h->wgred.assign((size_t)3 * nwg, WgReduceEntry{});
    for (size_t k = 0; k < P.nodes.size(); ++k) {
        const Node& n = P.nodes[k];
        if (n.wg3_S > 0) {
            for (int mode = 0; mode < 3; ++mode) {
                WgReduceEntry& e = h->wgred[(size_t)mode * nwg + n.wg3_entry];
                e.part = n.wg3_part; e.dst = P.convs[n.conv].w; e.numel = (int)P.wg3_numel(n);
                e.taps = n.type == N_STEM_CONV ? 1 : n.taps; e.pad_ = 0;
                e.S = wg3_active(P, n, mode == 1 ? 2 : (mode == 2 ? 1 : 0)) ? (mode == 1 ? n.wg3_S16 : n.wg3_S) : 0;
                if (mode == 0 && h->fused_S[k] > 0) e.S = h->fused_S[k];      // one partial tile per row block of the fused launch
            }
        }
    }
    hipStream_t s = (hipStream_t)stream;
    HIPCHK(hipMemcpyAsync(h->ws + P.off_wgred_tab, h->wgred.data(), h->wgred.size() * sizeof(WgReduceEntry), hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(h->ws + P.off_repack_tab, h->repack.data(), h->repack.size() * sizeof(RepackEntry), hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(h->ws + P.off_runstat_tab, h->runstat.data(), h->runstat.size() * sizeof(RunStatEntry), hipMemcpyHostToDevice, s));
    HIPCHK(hipStreamSynchronize(s));
    if (training && !h->side) {
        h->use_side = getenv("CUNET_NO_SIDE_STREAM") ? 0 : 1;
        // HIP multiplexes streams onto a few hardware queues; a same-priority side stream can land on the queue of the
        // caller's stream and then nothing overlaps (seen once RCCL had created its own streams).  A different
        // priority gets its own queue; the weight gradients are off the critical path, so: lowest priority.
        {
            int least = 0, greatest = 0;
            HIPCHK(hipDeviceGetStreamPriorityRange(&least, &greatest));
            const int mode = tune_int("CUNET_SIDE_PRIO", 1);
            if (mode == 0) HIPCHK(hipStreamCreateWithFlags(&h->side, hipStreamNonBlocking));
            else HIPCHK(hipStreamCreateWithPriority(&h->side, hipStreamNonBlocking, mode == 1 ? least : greatest));
        }
        h->fork_ev.resize(P.nodes.size());
        // Hand-over events order two streams of ONE device and are never inspected by the host (the caller synchronises its own stream):
        // CUNET_FORK_FLAGS bit 0 = device-scope release, bit 1 = no system-scope fence when the event completes, bit 2 = the same for the
        // events the caller's stream waits on (done / join / reduce / bucket).
        const int ff = tune_int("CUNET_FORK_FLAGS", 0);
        const unsigned scope_flags = ((ff & 1) ? hipEventReleaseToDevice : 0) | ((ff & 2) ? hipEventDisableSystemFence : 0);
        const unsigned fork_flags = hipEventDisableTiming | scope_flags;
        const unsigned back_flags = hipEventDisableTiming | ((ff & 4) ? scope_flags : 0);
        for (auto& e : h->fork_ev) HIPCHK(hipEventCreateWithFlags(&e, fork_flags));
        h->done_ev.resize(P.nodes.size());
        for (auto& e : h->done_ev) HIPCHK(hipEventCreateWithFlags(&e, back_flags));
        HIPCHK(hipEventCreateWithFlags(&h->join_ev, back_flags));
        for (auto& e : h->red_ev) HIPCHK(hipEventCreateWithFlags(&e, back_flags));
        HIPCHK(hipEventCreateWithFlags(&h->bucket_ev, back_flags));
    }
    h->fwd_training_done = 0; h->loss_done = 0;
    return CUNET_OK;
}

int cunet_set_quant_input(cunet_plan_t* h, int bits_i, const char* const* ternary_convs, int n_ternary) {
    if (!h || (bits_i != 0 && (bits_i < 3 || bits_i > 15)) || n_ternary < 0 || (n_ternary > 0 && !ternary_convs))
        return fail(CUNET_ERR_INVALID, "cunet_set_quant_input: bits_i must be 0 or 3..15");
    Plan& P = h->plan;
    h->qin_bits = bits_i;
    h->node_qin.assign(P.nodes.size(), 0);
    h->node_tern.assign(P.nodes.size(), 0);
    h->ternpack.clear();
    int count = 0;
    if (bits_i > 0) {
        for (size_t k = 0; k < P.nodes.size(); ++k) {
            const Node& n = P.nodes[k];
            if (n.type != N_CONV || !(n.taps == 9 || n.head >= 0)) continue;
            h->node_qin[k] = bits_i;
            const ConvInfo& c = P.convs[n.conv];
            if (c.tern < 0) continue;
            bool tern = false;
            for (int i = 0; i < n_ternary && !tern; ++i) tern = ternary_convs[i] && c.name == ternary_convs[i];
            if (!tern) continue;
            h->node_tern[k] = 1;
            TernPackEntry e{};
            e.src = c.w; e.dst = c.tern; e.O = c.Cout; e.C = c.Cin; e.taps = c.taps; e.Opad = round_up(c.Cout, 64);
            h->ternpack.push_back(e);
            ++count;
        }
    }
    h->ternpack_dirty = 1;
    return count;
}

int cunet_set_popcount_live(cunet_plan_t* h, int live) {
    if (!h) return fail(CUNET_ERR_INVALID, "null argument");
    h->tern_live = live ? 1 : 0;
    return CUNET_OK;
}

}  // extern "C"

// ---- executor helpers -----------------------------------------------------------------------
namespace {

struct Exec {
    cunet_plan* h;
    Plan& P;
    float* wsf;        // float region
    double* zero;      // fp64 region
    unsigned short* a16 = nullptr;   // bf16 activation arena when the last training forward was cunet_forward_bf16
    int xmode = 0;                   // the xbf16 value for kernel arguments: 0 fp32, 1 bf16 x, 2 bf16 x and bf16 gradient tensors
                                     // (a bf16 gradient tensor occupies the first half of its fp32 slot: same pointer)
    explicit Exec(cunet_plan* hh) : h(hh), P(hh->plan) {
        wsf = reinterpret_cast<float*>(h->ws + P.off_floats);
        zero = reinterpret_cast<double*>(h->ws + P.off_zero);
        if (h->fwd_training_done >= 2) {
            a16 = reinterpret_cast<unsigned short*>(h->ws + P.off_bf16_train);
            xmode = h->fwd_training_done == 3 ? 2 : 1;
        }
    }
    // activations as the backward kernels read them: fp32, or bf16 behind a float-typed pointer (xbf16 = 1 in the args)
    const float* xact(int t) const { return a16 ? reinterpret_cast<const float*>(a16 + P.tensors[t].act) : act(t); }
    float* act(int t) const { return wsf + P.tensors[t].act; }
    float* grad(int t) const { return wsf + P.tensors[t].grad; }
    double* stats(int t) const { return P.tensors[t].stats >= 0 ? zero + P.tensors[t].stats : nullptr; }

    int fill_segs(const Node& n, Seg* segs) const {
        int choff = 0;
        for (size_t i = 0; i < n.segs.size(); ++i) {
            const TensorInfo& t = P.tensors[n.segs[i].tensor];
            Seg& s = segs[i];
            s.x = xact(n.segs[i].tensor);
            s.gx = h->bound_training ? grad(n.segs[i].tensor) : nullptr;
            s.stats = stats(n.segs[i].tensor);
            s.count = (double)t.rows();
            s.C = t.C; s.ld = t.ld; s.ups = n.segs[i].ups; s.pad0_ = 0; s.choff = choff; s.pad_ = 0;
            choff += t.C;
        }
        return (int)n.segs.size();
    }
};

}  // namespace

// Leading dimension (elements) of the gradient tensor of heat-map head `n`.  fp32 gradients: the tensor's own ld.  bf16 gradient
// tensors: the K = class_num channels padded with zeros to the 32-multiple the bf16 MFMA data gradient contracts over (68 -> 96), when
// that fits the tensor's fp32 slot (2 * ld bf16 elements): the head's data gradient then runs on dgrad_bf16_kernel instead of the fp32
// kernel with its predicated 4-byte loads (84 us per head at 64 x 64, bs 24).
static int head_grad_ld(const Plan& P, const Node& n, int grad_bf16) {
    const TensorInfo& o = P.tensors[n.out];
    return (grad_bf16 && n.head >= 0) ? o.gld16 : o.ld;
}

// Heat-map head with the pixelwise MSE fused into its epilogue (cunet_loss_mse_fused): target staged in the workspace (NHWC),
// d(loss)/d(out) into the head's gradient tensor, the squared error into the loss accumulator.
static void set_fused_mse(cunet_plan* h, const Exec& E, const Node& n, ConvArgs& a) {
    Plan& P = h->plan;
    const TensorInfo& o = P.tensors[n.out];
    a.mse_tgt = E.wsf + P.target_off;
    a.mse_dout = E.grad(n.out);
    a.mse_acc = E.zero + P.loss_acc;
    a.mse_inv = 1.0 / ((double)o.rows() * o.C);
    a.mse_gbf16 = 0;
    a.mse_ldd = o.ld;
}

// Gradient of tensor `t`: one gather over the dz slices of the conv nodes that read it (all of them have
// run their data-gradient kernel: they come later in the forward order).  `only_node` >= 0 restricts the
// gather to one consumer (node-local debugging / tests).
// the launches of that gather, in order: plain consumers first, then the consumers behind the up-sample map, at most MAXGSRC per launch,
// every launch after the first accumulating
static void gather_launches(cunet_plan* h, int t, int only_node, std::vector<GradGatherArgs>& out) {
    Exec E(h);
    Plan& P = h->plan;
    const TensorInfo& ti = P.tensors[t];
    out.clear();
    for (int ups = 0; ups < 2; ++ups) {
        GradGatherArgs a{};
        auto flush = [&]() {
            if (a.nsrc == 0) return;
            a.accumulate = out.empty() ? 0 : 1;
            a.x = E.xact(t); a.xbf16 = E.xmode; a.gx = E.grad(t); a.stats = E.stats(t); a.count = (double)ti.rows();
            a.C = ti.C; a.ld = ti.ld; a.rows = (int)ti.rows(); a.H = ti.H; a.W = ti.W;
            out.push_back(a);
            a = GradGatherArgs{};
        };
        for (int i = 0; i < ti.ccount; ++i) {
            const Contrib& c = P.contribs[ti.cfirst + i];
            const Node& n = P.nodes[c.node];
            if (only_node >= 0 && c.node != only_node) continue;
            if (n.segs[c.seg].ups != ups) continue;
            int choff = 0;
            for (int j = 0; j < c.seg; ++j) choff += P.tensors[n.segs[j].tensor].C;
            GradSrc& g = a.src[a.nsrc++];
            g.dz = E.wsf + n.dz; g.red = E.zero + n.red; g.gamma = h->params + P.bns[n.bn].gamma;
            g.lddz = n.Ccat; g.choff = choff; g.ups = ups; g.pad_ = 0;
            if (a.nsrc == MAXGSRC) flush();
        }
        flush();
    }
}

static double gather_bytes(const GradGatherArgs& a) {
    return 4.0 * (double)a.rows * a.C * (2.0 + a.accumulate + a.nsrc * (a.src[0].ups ? 4.0 : 1.0));
}

static int gather_tensor_grad(cunet_plan* h, int t, int only_node, hipStream_t s) {
    std::vector<GradGatherArgs> ls;
    gather_launches(h, t, only_node, ls);
    for (const GradGatherArgs& a : ls)
        PROF(PC_APPLY, 0.0, gather_bytes(a), launch_grad_gather(a, h->num_cus, s));
    return CUNET_OK;
}

static PoolArgs pool_bwd_args(cunet_plan* h, Exec& E, const Node& n) {
    Plan& P = h->plan;
    const int tin = n.segs[0].tensor;
    const TensorInfo& ti = P.tensors[tin];
    PoolArgs a{};
    a.x = E.xact(tin); a.gy = E.grad(n.out); a.gx = E.grad(tin);
    a.xbf16 = E.xmode;
    a.N = ti.N; a.H = ti.H; a.W = ti.W; a.C = ti.C;
    return a;
}

// Backward of pool node `kp` together with the gather of the skip adapter in front of it (node kp - 1; planner option fuse_pool_gather):
// gather(pool output) -> pool backward and gather(skip adapter output) in ONE launch (gather_pool_pair_kernel).  1: launched (the caller
// must not gather the skip adapter's output again); 0: the pattern or the shapes do not fit, nothing launched; < 0: error in rc_out.
static int bwd_pool_with_skip_gather(cunet_plan* h, int kp, hipStream_t s, int& rc_out) {
    rc_out = CUNET_OK;
    Plan& P = h->plan;
    if (!P.opts.fuse_pool_gather || kp < 1) return 0;
    const Node& np = P.nodes[kp];
    const Node& ns = P.nodes[kp - 1];
    if (np.type != N_POOL || ns.type != N_CONV || ns.bucket != np.bucket || ns.name.find(".adapters_skip.") == std::string::npos) return 0;
    if (P.tensors[np.out].ccount <= 0 || P.tensors[ns.out].ccount <= 0 || P.tensors[np.segs[0].tensor].ld != P.tensors[np.segs[0].tensor].C) return 0;
    std::vector<GradGatherArgs> la, lb;
    gather_launches(h, np.out, -1, la);
    gather_launches(h, ns.out, -1, lb);
    if (la.size() != 1 || lb.size() != 1) return 0;
    Exec E(h);
    const PoolArgs pa = pool_bwd_args(h, E, np);
    int slot_;
    hipError_t e = prof_begin(h, PC_APPLY, s, slot_);
    if (e == hipSuccess) e = launch_gather_pool_pair(la[0], lb[0], pa, h->num_cus, s);
    if (e == hipErrorNotSupported) { prof_cancel(h, slot_); return 0; }
    const TensorInfo& ti = P.tensors[np.segs[0].tensor];
    if (e == hipSuccess) e = prof_end(h, slot_, 0.0, gather_bytes(la[0]) + gather_bytes(lb[0]) + 4.0 * 2.0 * (double)ti.rows() * ti.C, s);
    if (e != hipSuccess) { rc_out = fail(CUNET_ERR_HIP, std::string("gather + pool backward: ") + hipGetErrorString(e)); return -1; }
    return 1;
}

// dgamma / dbeta of the conv nodes [k0, k1) with bucket == `bucket` (or every bucket if < 0)
static int bn_param_grads(cunet_plan* h, int k0, int k1, int bucket, hipStream_t s) {
    Exec E(h);
    Plan& P = h->plan;
    BnParamGradArgs a{};
    for (int k = k0; k < k1; ++k) {
        const Node& n = P.nodes[k];
        if (n.type != N_CONV || (bucket >= 0 && n.bucket != bucket)) continue;
        const BnInfo& b = P.bns[n.bn];
        auto& e = a.e[a.n++];
        e.red = E.zero + n.red; e.dgamma = h->grads + b.gamma; e.dbeta = h->grads + b.beta; e.C = n.Ccat; e.pad_ = 0;
        if (a.n == MAXBNG) { HIPCHK(launch_bn_param_grad(a, s)); a.n = 0; }
    }
    HIPCHK(launch_bn_param_grad(a, s));
    return CUNET_OK;
}

// Sums the partial tiles of the wgrad3 nodes [first, first + count) of the reduce table into the gradient arena, on the
// stream that ran those weight gradients.
static int reduce_wgrad3(cunet_plan* h, int first, int count, int max_numel, hipStream_t s) {
    Exec E(h);
    if (count <= 0) return CUNET_OK;
    Plan& P = h->plan;
    const int nwg = P.n_wgred > 0 ? P.n_wgred : 1;
    const WgReduceEntry* tab = reinterpret_cast<const WgReduceEntry*>(h->ws + P.off_wgred_tab) + (E.xmode == 2 ? nwg : (E.xmode == 1 ? 2 * nwg : 0)) + first;
    PROF_ON(s, PC_MISC, 0.0, 0.0, launch_wgrad_reduce(tab, count, max_numel, E.wsf, h->grads, s));
    return CUNET_OK;
}

// Backward of one node.  BWD_MAIN: what the rest of backward waits for -- data gradient (+ReLU mask + BN reductions), pool /
// stem BN-pool backward -- on `s`.  BWD_WGRAD: the weight gradient, which only reads d(loss)/d(out) and activations and only
// writes dW (or partial tiles), on `ws`: the caller decides where that is and when it may start (fork_wgrads).
enum { BWD_MAIN = 1, BWD_WGRAD = 2 };
static bool node_has_wgrad(const Node& n) { return n.type == N_CONV || n.type == N_STEM_CONV; }

// arguments of a conv node's data gradient (+ ReLU mask + BatchNorm reductions)
static ConvArgs dgrad_args(cunet_plan* h, Exec& E, const Node& n, int node_index) {
    Plan& P = h->plan;
    const ConvInfo& c = P.convs[n.conv];
    const BnInfo& b = P.bns[n.bn];
    const TensorInfo& o = P.tensors[n.out];
    ConvArgs a{};
    a.nseg = E.fill_segs(n, a.seg); a.Ccat = n.Ccat;
    a.gamma = h->params + b.gamma; a.beta = h->params + b.beta;
    a.rmean = h->buffers + b.rmean; a.rvar = h->buffers + b.rvar;
    a.training = 1;
    a.xbf16 = E.xmode;
    a.qin_bits = h->qin_bits ? h->node_qin[node_index] : 0;
    a.a = E.grad(n.out); a.lda = head_grad_ld(P, n, E.xmode == 2);      // (a padded bf16 head gradient: see head_grad_ld)
    a.K = c.Cout; a.taps = c.taps; a.wB = E.wsf + c.wB; a.Kpad = c.KpadB; a.Npad = c.NpadB;
    a.y = n.dz >= 0 ? E.wsf + n.dz : nullptr; a.ldy = n.Ccat; a.Nout = n.Ccat; a.ystats = E.zero + n.red;
    a.M = (int)o.rows(); a.H = o.H; a.W = o.W;
    a.dgrad_nt = P.opts.dgrad_nt;
    a.dgrad3_nt = P.opts.dgrad3_nt;
    a.dgrad3_ring = P.opts.dgrad3_ring;
    a.dgrad_prefetch = P.opts.dgrad_prefetch;
    // (default -1: with the split contraction the 64 x 64 launches -- 3072 row tiles at batch 24 -- take the row-tile kernel, which cuts dY
    // into its bf16 pieces once per workgroup instead of once per column slice: +0.7 ... 1.7 % on the CU-Net-2 step; on the fp32 pipe never)
    a.dgrad_rows = P.opts.dgrad_rows >= 0 ? P.opts.dgrad_rows : (P.opts.f32_split ? 3072 : 0);
    a.dgrad_rows_v = P.opts.dgrad_rows_v;
    a.split = P.opts.f32_split;
    // fp32 gradient tensors: this launch also computes the node's weight gradient (partial tiles; the bucket's reduce sums them)
    a.wg_part = (E.xmode == 0 && !h->fused_S.empty() && h->fused_S[node_index] > 0) ? E.wsf + n.wg3_part : nullptr;
    if (h->fz_node == node_index) {      // the one consumer's BatchNorm backward applied on the operand load (see fz_eligible)
        const Contrib& cb = P.contribs[o.cfirst];
        const Node& nc = P.nodes[cb.node];
        int choff = 0;
        for (int j = 0; j < cb.seg; ++j) choff += P.tensors[nc.segs[j].tensor].C;
        a.fz_dz = E.wsf + nc.dz; a.fz_red = E.zero + nc.red; a.fz_gamma = h->params + P.bns[nc.bn].gamma;
        a.fz_lddz = nc.Ccat; a.fz_choff = choff;
        a.fz_x = E.xact(n.out); a.fz_ldx = o.ld; a.fz_stats = E.stats(n.out); a.fz_count = (double)o.rows();
    }
    return a;
}

// May node k's data gradient assemble d(loss)/d(out) itself (ConvArgs::fz_*) instead of reading the tensor a gather launch wrote?  A 1x1
// conv over 128 output channels whose output has exactly ONE consumer, at the same resolution -- the bottleneck output of a dense layer
// (models/cu_net.py:43-48: conv1 -> norm2 -> relu -> conv2) -- with bf16 gradient tensors (the kernels that implement it so far).
static bool fz_eligible(const cunet_plan* h, int k, int xmode) {
    const Plan& P = h->plan;
    const Node& n = P.nodes[k];
    if (!P.opts.fuse_z_gather || xmode != 2 || n.type != N_CONV || n.taps != 1 || n.head >= 0) return false;
    const TensorInfo& o = P.tensors[n.out];
    if (o.ccount != 1 || o.C != 128 || o.ld != 128 || P.convs[n.conv].Cout != 128 || o.stats < 0) return false;
    const Contrib& cb = P.contribs[o.cfirst];
    const Node& nc = P.nodes[cb.node];
    return nc.type == N_CONV && nc.dz >= 0 && nc.segs[cb.seg].ups == 0 && nc.segs[cb.seg].tensor == n.out;
}
constexpr int CUNET_FZ_FALLBACK = 1;      // bwd_node: the fused launch does not exist for this shape -- nothing launched; gather, then launch plainly

// Is node k's weight gradient computed by its data-gradient launch in the current pass?
static bool wgrad_is_fused(const cunet_plan* h, int k, int xmode) { return xmode == 0 && !h->fused_S.empty() && h->fused_S[k] > 0; }

// bind(): which fp32 1x1 nodes fuse their weight gradient into the data gradient, and the partial tiles each such launch writes (the row
// blocks of its grid -- a function of the shape, the CU count and whether the node is launched as half of an adapter pair; it has to
// fit the slice of the partial region the planner reserved, Node::wg3_cap).
static int plan_fused_wgrads(cunet_plan* h) {
    Plan& P = h->plan;
    if (!P.opts.fuse_wgrad) return CUNET_OK;
    const int saved = h->fwd_training_done;
    h->fwd_training_done = 0;                                  // (dgrad_args of the fp32 mode)
    Exec E(h);
    std::vector<int> tmp(P.nodes.size(), 0);
    h->fused_S.assign(P.nodes.size(), 0);
    for (size_t k = 0; k < P.nodes.size(); ++k) {
        const Node& n = P.nodes[k];
        if (n.type != N_CONV || !n.fuse_ok || n.taps != 1 || n.head >= 0 || n.wg3_S <= 0 || !wg3_active(P, n, 0)) continue;
        const bool first_of_pair = n.pair && k + 1 < P.nodes.size() && P.nodes[k + 1].bucket == n.bucket;
        const bool second_of_pair = k >= 1 && P.nodes[k - 1].pair && P.nodes[k - 1].bucket == n.bucket;
        if (second_of_pair) continue;                          // (decided with its partner)
        ConvArgs a = dgrad_args(h, E, n, (int)k);
        if (first_of_pair) {
            const Node& n2 = P.nodes[k + 1];
            const ConvArgs a2 = dgrad_args(h, E, n2, (int)k + 1);
            const bool both = n2.fuse_ok && n2.wg3_S > 0 && n2.taps == 1 && wg3_active(P, n2, 0) && conv_pairable(a, a2);
            const int S = both ? conv_fused_wgrad_splits(a, true, h->num_cus) : 0;
            if (S > 0 && S <= n.wg3_cap && S <= n2.wg3_cap) tmp[k] = tmp[k + 1] = S;
            continue;                                          // (a pair is fused together or not at all: one launch)
        }
        const int S = conv_fused_wgrad_splits(a, false, h->num_cus);
        if (S > 0 && S <= n.wg3_cap) tmp[k] = S;
    }
    h->fused_S = tmp;
    h->fwd_training_done = saved;
    return CUNET_OK;
}

// the same for dgrad_bf16_kernel (bf16 gradient tensors): the bf16 backward operand
static ConvArgs dgrad_args_bf16(cunet_plan* h, Exec& E, const Node& n, const ConvArgs& a) {
    const ConvInfo& c = h->plan.convs[n.conv];
    ConvArgs b16 = a;
    b16.wB = reinterpret_cast<const float*>(E.a16 + c.wB);
    if (n.head >= 0) b16.K = c.KpadB;             // the zero-padded channels are contracted too (zero weights behind them)
    return b16;
}

// Data gradients of an adapter pair (nodes k0 = ahead and k1 = skip, Node::pair) in one launch.  1: launched; 0: no pair kernel for
// this shape / storage (nothing launched, the caller runs them one by one); < 0: error.
static int bwd_dgrad_pair(cunet_plan* h, int k0, int k1, hipStream_t s, int& rc_out) {
    Exec E(h);
    Plan& P = h->plan;
    const Node& n0 = P.nodes[k0];
    const Node& n1 = P.nodes[k1];
    const ConvArgs a0 = dgrad_args(h, E, n0, k0), a1 = dgrad_args(h, E, n1, k1);
    static const int dg16 = tune_int("CUNET_NO_DGRAD_BF16", 0) ? 0 : 1;
    const bool on16 = E.xmode == 2 && dg16;
    int slot_;
    rc_out = CUNET_OK;
#define PAIRCHK(expr) do { hipError_t e__ = (expr); if (e__ != hipSuccess) { rc_out = fail(CUNET_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e__)); return -1; } } while (0)
    PAIRCHK(prof_begin(h, on16 ? PC_C1D16 : PC_C1D, s, slot_));
    hipError_t e;
    if (on16) e = launch_dgrad_bf16_pair(dgrad_args_bf16(h, E, n0, a0), dgrad_args_bf16(h, E, n1, a1), h->num_cus, s);
    else e = launch_conv_pair(a0, a1, LD_PLAIN, EP_BWD, h->num_cus, s);
    if (e == hipErrorNotSupported || e == hipErrorInvalidValue) {
        prof_cancel(h, slot_);
        if (a0.wg_part || a1.wg_part) {        // (the reduce table was sized for the pair's geometry at bind: no silent fall-back)
            rc_out = fail(CUNET_ERR_STATE, "internal: the fused data + weight gradient of an adapter pair has no pair launch");
            return -1;
        }
        return 0;
    }
    PAIRCHK(e);
    PAIRCHK(prof_end(h, slot_, 2.0 * 2.0 * a0.M * a0.K * a0.Nout, 2.0 * (on16 ? 2.0 : 4.0) * (double)a0.M * (a0.K + 2.0 * a0.Nout), s));
#undef PAIRCHK
    return 1;
}

static int bwd_node(cunet_plan* h, const Node& n, int node_index, hipStream_t s, hipStream_t ws, int parts) {
    Exec E(h);
    Plan& P = h->plan;
    const int cus = h->num_cus;
    const TensorInfo& o = P.tensors[n.out];
    if (n.type == N_CONV) {
        const ConvInfo& c = P.convs[n.conv];
        const BnInfo& b = P.bns[n.bn];
        if (parts & BWD_MAIN) {   // data gradient + ReLU mask + BN reductions
            const ConvArgs a = dgrad_args(h, E, n, node_index);
            const int gld = a.lda;
            static const int dg16 = tune_int("CUNET_NO_DGRAD_BF16", 0) ? 0 : 1;
            bool done = false;
            if (E.xmode == 2 && dg16 && (n.head < 0 || gld != o.ld)) {      // bf16 gradient tensors: bf16 MFMA data gradient where the shape allows
                const ConvArgs b16 = dgrad_args_bf16(h, E, n, a);
                int slot_;
                HIPCHK(prof_begin(h, c.taps == 9 ? PC_C3D16 : PC_C1D16, s, slot_));
                const hipError_t e = launch_dgrad_bf16(b16, cus, s);
                if (e == hipSuccess) {
                    HIPCHK(prof_end(h, slot_, 2.0 * a.M * a.K * a.Nout * a.taps, 2.0 * (double)a.M * (a.K + 2.0 * a.Nout), s));
                    done = true;
                } else if (a.fz_dz != nullptr && (e == hipErrorNotSupported || e == hipErrorInvalidValue)) {
                    prof_cancel(h, slot_);
                    return CUNET_FZ_FALLBACK;
                } else if (e != hipErrorInvalidValue) {
                    HIPCHK(e);
                } else {
                    HIPCHK(prof_end(h, slot_, 0.0, 0.0, s));
                }
            }
            if (!done && a.fz_dz != nullptr) return CUNET_FZ_FALLBACK;      // (the fp32 kernels read the gathered tensor)
            if (!done)
                PROF(c.taps == 9 ? PC_C3D : PC_C1D, 2.0 * a.M * a.K * a.Nout * a.taps, 4.0 * (double)a.M * (a.K + 2.0 * a.Nout),
                     launch_conv(a, c.taps == 9 ? LD_PLAIN3 : LD_PLAIN, EP_BWD, cus, s));
        }
        if ((parts & BWD_WGRAD) && !wgrad_is_fused(h, node_index, E.xmode)) {   // weight gradient (a fused node's came with its data gradient)
            WgradArgs w{};
            w.dy = E.grad(n.out); w.lddy = head_grad_ld(P, n, E.xmode == 2); w.Cout = c.Cout;
            w.nseg = E.fill_segs(n, w.seg); w.Ccat = n.Ccat;
            w.gamma = h->params + b.gamma; w.beta = h->params + b.beta;
            w.taps = c.taps; w.M = (int)o.rows(); w.H = o.H; w.W = o.W;
            w.dw = h->grads + c.w;
            w.xbf16 = E.xmode;
            w.qin_bits = h->qin_bits ? h->node_qin[node_index] : 0;
            w.split = E.xmode == 0 ? P.opts.f32_split : 0;
            w.split_planes = P.opts.wgrad_split_planes;
            w.bf16_dma = P.opts.wgrad_bf16_dma;      // (the plan's snapshot, like every other option; cunet_debug_set_plan_option flips it on a live plan for the bit-identity test)
            if (c.taps == 9 && wg3_active(P, n, E.xmode) && wgrad3_3x3_supported(w)) {
                // LDS ring of activated image rows, partial tiles [split][tap][n][c] (reduced + transposed per bucket)
                const bool on16 = wgrad3_3x3_on_bf16_mfma(w);
                PROF_ON(ws, on16 ? PC_C3W16 : PC_C3W, 2.0 * w.M * w.Cout * w.Ccat * 9, (on16 ? 2.0 : 4.0) * (double)w.M * (w.Cout + w.Ccat),
                        launch_wgrad3_3x3(w, E.wsf + n.wg3_part, E.xmode == 2 ? n.wg3_S16 : n.wg3_S, E.xmode == 2 ? n.wg3_rows16 : n.wg3_rows, ws));
            } else if (c.taps == 1 && wg3_active(P, n, E.xmode) && wgrad3_supported(w)) {
                // LDS-staged, atomics-free: partial tiles now, summed into the arena by the bucket's reduce (reduce_wgrad3)
                PROF_ON(ws, E.xmode == 2 ? PC_C1W16 : PC_C1W, 2.0 * w.M * w.Cout * w.Ccat, (E.xmode == 2 ? 2.0 : 4.0) * (double)w.M * (w.Cout + w.Ccat),
                        launch_wgrad3(w, E.wsf + n.wg3_part, E.xmode == 2 ? n.wg3_S16 : n.wg3_S, E.xmode == 2 ? n.wg3_rows16 : n.wg3_rows, ws));
            } else {
                PROF_ON(ws, c.taps == 9 ? PC_C3W : PC_C1W, 2.0 * w.M * w.Cout * w.Ccat * w.taps, 4.0 * (double)w.M * (w.Cout + w.Ccat),
                        launch_wgrad(w, c.taps == 9 ? WGL_3X3 : WGL_SEG, cus, ws));
            }
        }
    } else if (n.type == N_POOL) {
        if (!(parts & BWD_MAIN)) return CUNET_OK;
        const TensorInfo& ti = P.tensors[n.segs[0].tensor];
        const PoolArgs a = pool_bwd_args(h, E, n);
        PROF(PC_POOLB, 0.0, 4.0 * 2.25 * (double)ti.rows() * ti.C, launch_pool_bwd(a, cus, s));
    } else if (n.type == N_STEM_BNPOOL) {
        if (!(parts & BWD_MAIN)) return CUNET_OK;
        const int tin = n.segs[0].tensor;
        const TensorInfo& ti = P.tensors[tin];
        const BnInfo& b = P.bns[n.bn];
        PoolArgs a{};
        a.x = E.act(tin); a.gy = E.grad(n.out); a.gx = E.grad(tin);
        a.xstats = E.stats(tin); a.count = (double)ti.rows();
        a.gamma = h->params + b.gamma; a.beta = h->params + b.beta;
        a.N = ti.N; a.H = ti.H; a.W = ti.W; a.C = ti.C; a.training = 1;
        a.red = E.zero + n.red;
        a.xbf16 = E.xmode == 2 ? 2 : 0;          // x (the stem conv output) is fp32 in every mode; gy follows the gradient storage
        PROF(PC_STEMBPB, 0.0, 4.0 * 1.25 * (double)ti.rows() * ti.C, launch_stem_bwd(a, 0, nullptr, nullptr, cus, s));
        if (h->stem_fused_now)      // (planner option stem_fuse_dz: the stem's weight gradient derives dz itself; only dgamma / dbeta are left to do here)
            HIPCHK(launch_stem_bwd(a, 2, h->grads + b.gamma, h->grads + b.beta, cus, s));
        else
            PROF(PC_STEMBPB, 0.0, 4.0 * 2.25 * (double)ti.rows() * ti.C, launch_stem_bwd(a, 1, h->grads + b.gamma, h->grads + b.beta, cus, s));
    } else {  // N_STEM_CONV: weight gradient only (the image needs no gradient)
        if (!(parts & BWD_WGRAD)) return CUNET_OK;
        const ConvInfo& c = P.convs[n.conv];
        WgradArgs w{};
        w.dy = E.grad(n.out); w.lddy = o.ld; w.Cout = c.Cout;
        w.nseg = 0; w.Ccat = c.Cin; w.taps = 1;
        w.M = (int)o.rows(); w.H = o.H; w.W = o.W;
        w.dw = h->grads + c.w;
        w.img = h->last_x; w.IH = P.cfg.height; w.IW = P.cfg.width;
        w.split = (P.opts.f32_split && P.opts.stem_wgrad_split) ? 1 : 0;      // (fp32 image and fp32 dY in every storage mode)
        w.split_planes = P.opts.stem_wgrad_planes;                            // (round 6: both operands cut once on their way into LDS)
        if (wg3_active(P, n, E.xmode)) {      // (an unsupported shape is an error here: the reduce table already expects partial tiles)
            if (h->stem_fused_now) {           // d(loss)/d(conv0 output) is computed inside the kernel (WgradArgs::sx ...)
                const Node& nb = P.nodes[node_index + 1];       // the stem's BatchNorm-ReLU-pool node
                const BnInfo& sb = P.bns[nb.bn];
                w.sx = E.act(n.out); w.sgy = E.grad(nb.out); w.sstats = E.stats(n.out); w.sred = E.zero + nb.red; w.scount = (double)o.rows();
                w.gamma = h->params + sb.gamma; w.beta = h->params + sb.beta;
            }
            PROF_ON(ws, PC_STEMW, 2.0 * w.M * w.Cout * w.Ccat, 4.0 * (double)w.M * w.Cout,
                    launch_wgrad3_stem(w, E.wsf + n.wg3_part, n.wg3_wpi, n.wg3_rows, ws));
        } else {
            PROF_ON(ws, PC_STEMW, 2.0 * w.M * w.Cout * w.Ccat, 4.0 * (double)w.M * w.Cout, launch_wgrad(w, WGL_STEM, cus, ws));
        }
    }
    return CUNET_OK;
}

// Weight gradients of the nodes in `pending` (their d(loss)/d(out) has been enqueued on `s`): ONE event on `s`, the side stream
// waits for it, then all of them go to the side stream.  An event record is a marker packet in the caller's queue and the kernel
// behind it waits for the marker to retire: 6-7 us of bubble on the critical path per record (rocprofv3 traces of round 3: 0.7 ms
// per CU-Net-2 step and 3.2 ms per CU-Net-8 bf16 step when every node recorded its own) -- hence groups.
static int fork_wgrads(cunet_plan* h, std::vector<int>& pending, hipStream_t s) {
    if (pending.empty()) return CUNET_OK;
    Plan& P = h->plan;
    hipStream_t ws = s;
    if (h->use_side && h->side) {
        HIPCHK(hipEventRecord(h->fork_ev[pending.back()], s));
        HIPCHK(hipStreamWaitEvent(h->side, h->fork_ev[pending.back()], 0));
        ws = h->side;
    }
    for (int k : pending) {
        const int rc = bwd_node(h, P.nodes[k], k, s, ws, BWD_WGRAD);
        if (rc != CUNET_OK) return rc;
    }
    pending.clear();
    return CUNET_OK;
}

// arguments of a conv node's fp32 forward ([concat -> BatchNorm -> ReLU (-> QuanInput)] -> conv)
static ConvArgs conv_fwd_args(cunet_plan* h, Exec& E, const Node& n, int node_index, int training) {
    Plan& P = h->plan;
    const ConvInfo& c = P.convs[n.conv];
    const BnInfo& b = P.bns[n.bn];
    const TensorInfo& o = P.tensors[n.out];
    ConvArgs a{};
    a.nseg = E.fill_segs(n, a.seg); a.Ccat = n.Ccat;
    a.gamma = h->params + b.gamma; a.beta = h->params + b.beta;
    a.rmean = h->buffers + b.rmean; a.rvar = h->buffers + b.rvar;
    a.training = training;
    a.K = n.Ccat; a.taps = c.taps; a.wB = E.wsf + c.wF; a.Kpad = c.KpadF; a.Npad = c.NpadF;
    a.y = E.act(n.out); a.ldy = o.ld; a.Nout = c.Cout; a.ystats = training ? E.stats(n.out) : nullptr;
    a.M = (int)o.rows(); a.H = o.H; a.W = o.W;
    a.qin_bits = h->qin_bits ? h->node_qin[node_index] : 0;
    a.ring_min_rows = P.opts.conv3x3_ring_min_rows > 0 ? P.opts.conv3x3_ring_min_rows : 1;
    a.split = P.opts.f32_split;
    return a;
}

// the same for the bf16 forward: inputs, weights and (except heads) the output in the bf16 arena
static ConvArgs conv_fwd_args_bf16(cunet_plan* h, Exec& E, const Node& n, unsigned short* a16, int training) {
    Plan& P = h->plan;
    const ConvInfo& c = P.convs[n.conv];
    const BnInfo& b = P.bns[n.bn];
    const TensorInfo& o = P.tensors[n.out];
    ConvArgs a{};
    a.nseg = E.fill_segs(n, a.seg); a.Ccat = n.Ccat;
    for (int i = 0; i < a.nseg; ++i)       // the bf16 copies of the inputs
        a.seg[i].x = reinterpret_cast<const float*>(a16 + P.tensors[n.segs[i].tensor].act);
    a.gamma = h->params + b.gamma; a.beta = h->params + b.beta;
    a.rmean = h->buffers + b.rmean; a.rvar = h->buffers + b.rvar;
    a.training = training ? 1 : 0;
    a.K = n.Ccat; a.taps = c.taps; a.wB = reinterpret_cast<const float*>(a16 + c.wF); a.Kpad = c.KpadF; a.Npad = c.NpadF;
    const int is_head = n.head >= 0;
    a.y = is_head ? E.act(n.out) : reinterpret_cast<float*>(a16 + o.act);
    a.ldy = o.ld; a.Nout = c.Cout; a.ystats = (training && !is_head) ? E.stats(n.out) : nullptr;
    a.M = (int)o.rows(); a.H = o.H; a.W = o.W;
    a.ring_min_rows = P.opts.conv3x3_ring_min_rows > 0 ? P.opts.conv3x3_ring_min_rows : 1;
    return a;
}

extern "C" {

int cunet_forward(cunet_plan_t* h, const float* x, float* const* heat, int training, void* stream) {
    if (!h || !x) return fail(CUNET_ERR_INVALID, "null argument");
    if (!h->ws) return fail(CUNET_ERR_STATE, "cunet_bind has not been called");
    hipStream_t s = (hipStream_t)stream;
    h->fwd_training_done = 0;        // (an earlier bf16 forward must not redirect this pass's activation pointers)
    h->stem_fused_now = false;       // (cunet_debug_materialise: the reductions it would read are about to be cleared)
    Exec E(h);
    Plan& P = h->plan;
    const int cus = h->num_cus;
    HIPCHK(hipMemsetAsync(h->ws + P.off_zero, 0, (size_t)P.zero_bytes, s));
    HIPCHK(launch_repack(reinterpret_cast<const RepackEntry*>(h->ws + P.off_repack_tab), (int)P.convs.size(), h->params, E.wsf, s));
    if (h->qin_bits && h->tern_live && !h->ternpack.empty()) {          // bit masks of the convs that run on AND-popcount this pass
        if (h->ternpack_dirty) {
            HIPCHK(hipMemcpyAsync(h->ws + P.off_ternpack_tab, h->ternpack.data(), h->ternpack.size() * sizeof(TernPackEntry), hipMemcpyHostToDevice, s));
            HIPCHK(hipStreamSynchronize(s));             // (the host vector may change before an async copy has read it)
            h->ternpack_dirty = 0;
        }
        HIPCHK(launch_ternary_pack_all(reinterpret_cast<const TernPackEntry*>(h->ws + P.off_ternpack_tab), (int)h->ternpack.size(), h->params,
                                       reinterpret_cast<uint64_t*>(h->ws + P.off_tern), s));
    }
    const bool fork_fwd = training && h->use_side && h->side && !h->done_ev.empty();
    h->pending.assign(P.tensors.size(), -1);
    hipStream_t s_main = s;
    for (size_t ni = 0; ni < P.nodes.size(); ++ni) {
        const Node& n = P.nodes[ni];
        const TensorInfo& o = P.tensors[n.out];
        s = s_main;
        // join: an input produced on the side stream must be complete
        for (const SegRef& sr : n.segs) {
            const int pn = h->pending[sr.tensor];
            if (pn >= 0) { HIPCHK(hipStreamWaitEvent(s_main, h->done_ev[pn], 0)); h->pending[sr.tensor] = -1; }
        }
        // the ahead and the skip adapter of a down block read the same concat: one launch for both where the shape has a pair kernel
        if (n.pair && ni + 1 < P.nodes.size()) {
            const Node& n2 = P.nodes[ni + 1];
            const ConvArgs a0 = conv_fwd_args(h, E, n, (int)ni, training), a1 = conv_fwd_args(h, E, n2, (int)ni + 1, training);
            int slot_;
            HIPCHK(prof_begin(h, PC_C1F, s, slot_));
            const hipError_t e = launch_conv_pair(a0, a1, LD_SEG, EP_FWD, cus, s);
            if (e == hipSuccess) {
                HIPCHK(prof_end(h, slot_, 2.0 * 2.0 * a0.M * a0.K * a0.Nout, 2.0 * 4.0 * (double)a0.M * (a0.K + a0.Nout), s));
                ++ni;
                continue;
            }
            prof_cancel(h, slot_);
            // (no pair kernel for the shape, or the pair's configuration does not fit -- e.g. its LDS budget: the single launches below)
            if (e != hipErrorNotSupported && e != hipErrorInvalidValue) HIPCHK(e);
        }
        // fork: the skip adapter of a down block is consumed only on the way up (models/cu_net.py:257,267),
        // so it runs on the side stream next to the ahead adapter / pool / next block
        // ... and so does a heat-map head in a training pass: nothing in the forward reads its output (the loss is finalised after
        // the join below), and the side stream is idle in the forward
        // (never a node of the AND-popcount forward: every popcount site shares ONE bit-plane scratch, `off_planes`, reused site
        // after site in stream order -- a head on the side stream would still be reading its records while the next U-Net's 3x3
        // site rewrites them on the caller's stream)
        const bool popcount_node = n.type == N_CONV && h->qin_bits && h->tern_live && h->node_qin[ni] && h->node_tern[ni];
        // (round 6: never the LAST node of the list -- the last U-Net's head: nothing is left to run beside it, and the hand-over and the
        // join around it were ~20 us of idle GPU between the forward and the backward)
        const bool forked = fork_fwd && n.type == N_CONV && !popcount_node &&
                            ((o.W >= P.opts.fwd_fork_min_w && n.name.find(".adapters_skip.") != std::string::npos) ||
                             (n.head >= 0 && P.opts.heads_on_side && ni + 1 < P.nodes.size()));
        if (forked) {
            HIPCHK(hipEventRecord(h->fork_ev[ni], s_main));
            HIPCHK(hipStreamWaitEvent(h->side, h->fork_ev[ni], 0));
            s = h->side;
        }
        if (n.type == N_STEM_CONV) {
            const ConvInfo& c = P.convs[n.conv];
            ConvArgs a{};
            a.nseg = 0; a.Ccat = 0; a.training = training ? 1 : 0;
            a.K = c.Cin; a.taps = 1; a.wB = E.wsf + c.wF; a.Kpad = c.KpadF; a.Npad = c.NpadF;
            a.y = E.act(n.out); a.ldy = o.ld; a.Nout = c.Cout; a.ystats = training ? E.stats(n.out) : nullptr;
            a.M = (int)o.rows(); a.H = o.H; a.W = o.W;
            a.img = x; a.IH = P.cfg.height; a.IW = P.cfg.width;
            a.split = (P.opts.f32_split && P.opts.stem_split) ? 1 : 0;
            PROF(PC_STEMF, 2.0 * a.M * a.K * a.Nout, 4.0 * ((double)a.M * a.Nout + (double)o.N * 3 * a.IH * a.IW), launch_conv(a, LD_STEM, EP_FWD, cus, s));
        } else if (n.type == N_STEM_BNPOOL || n.type == N_POOL) {
            const int tin = n.segs[0].tensor;
            const TensorInfo& ti = P.tensors[tin];
            PoolArgs a{};
            a.x = E.act(tin); a.y = E.act(n.out); a.ystats = training ? E.stats(n.out) : nullptr;
            a.N = ti.N; a.H = ti.H; a.W = ti.W; a.C = ti.C; a.training = training;
            if (n.type == N_STEM_BNPOOL) {
                const BnInfo& b = P.bns[n.bn];
                a.xstats = E.stats(tin); a.count = (double)ti.rows();
                a.gamma = h->params + b.gamma; a.beta = h->params + b.beta;
                a.rmean = h->buffers + b.rmean; a.rvar = h->buffers + b.rvar;
                PROF(PC_STEMBPF, 0.0, 4.0 * 1.25 * (double)ti.rows() * ti.C, launch_pool_fwd(a, 1, cus, s));
            } else {
                PROF(PC_POOLF, 0.0, 4.0 * 1.25 * (double)ti.rows() * ti.C, launch_pool_fwd(a, 0, cus, s));
            }
        } else {  // N_CONV
            const ConvInfo& c = P.convs[n.conv];
            ConvArgs a = conv_fwd_args(h, E, n, (int)ni, training);
            const bool on_popcount = a.qin_bits && h->node_tern[ni] && h->tern_live;
            const bool fuse_mse = n.head >= 0 && training && h->fused_loss_out != nullptr;
            if (fuse_mse && !on_popcount) set_fused_mse(h, E, n, a);
            if (on_popcount) {
                // ternary weights x quantised activations: multiplier-free AND-popcount forward (one input tensor: the
                // bottleneck output for a 3x3 conv, the U-Net output for a head)
                const TensorInfo& ti = P.tensors[n.segs[0].tensor];
                const int64_t words = (int64_t)c.taps * ((c.Cin + 63) / 64) * round_up(c.Cout, 64);
                TernArgs t{};
                t.x = E.act(n.segs[0].tensor); t.ldx = ti.ld; t.scale = nullptr; t.shift = nullptr;
                t.wpos = reinterpret_cast<const uint64_t*>(h->ws + P.off_tern) + c.tern; t.wneg = t.wpos + words;
                t.y = E.act(n.out); t.ldy = o.ld;
                t.M = (int)o.rows(); t.H = o.H; t.W = o.W; t.C = c.Cin; t.O = c.Cout; t.Opad = round_up(c.Cout, 64); t.taps = c.taps; t.bits_i = a.qin_bits;
                t.xstats = E.stats(n.segs[0].tensor); t.count = (double)ti.rows();
                t.gamma = a.gamma; t.beta = a.beta; t.rmean = a.rmean; t.rvar = a.rvar; t.training = training ? 1 : 0;
                t.ystats = a.ystats;
                t.planes = reinterpret_cast<uint64_t*>(h->ws + P.off_planes);
                t.variant = P.opts.popcount_pixels;
                PROF(PC_TERN, 0.0, 4.0 * (double)a.M * (a.K + a.Nout), launch_ternary_conv(t, cus, s));
                if (fuse_mse)       // (the AND-popcount kernel has no loss epilogue: this head's MSE is its own launch)
                    HIPCHK(launch_mse(E.act(n.out), E.wsf + P.target_off, E.grad(n.out), E.zero + P.loss_acc, (long)o.rows(), o.C, o.ld, o.ld, 0, cus, s));
            } else {
                PROF(c.taps == 9 ? PC_C3F : PC_C1F, 2.0 * a.M * a.K * a.Nout * a.taps, 4.0 * (double)a.M * (a.K + a.Nout),
                     launch_conv(a, c.taps == 9 ? LD_3X3 : LD_SEG, EP_FWD, cus, s));
            }
        }
        if (forked) {
            HIPCHK(hipEventRecord(h->done_ev[ni], h->side));
            h->pending[n.out] = (int)ni;
        }
    }
    s = s_main;
    for (size_t t = 0; t < h->pending.size(); ++t)      // nothing may be left floating
        if (h->pending[t] >= 0) { HIPCHK(hipStreamWaitEvent(s, h->done_ev[h->pending[t]], 0)); h->pending[t] = -1; }
    if (training)
        HIPCHK(launch_running_update(reinterpret_cast<const RunStatEntry*>(h->ws + P.off_runstat_tab), P.n_runstat,
                                     E.zero, h->buffers, h->counters, 0, s));
    if (heat) {
        for (size_t i = 0; i < P.head_tensors.size(); ++i) {
            if (!heat[i]) continue;
            const TensorInfo& t = P.tensors[P.head_tensors[i]];
            HIPCHK(launch_transpose(E.act(P.head_tensors[i]), heat[i], t.N, t.C, t.H * t.W, t.ld, 0, s));
        }
    }
    h->fwd_training_done = training ? 1 : 0;
    h->loss_done = 0;
    if (training && h->fused_loss_out != nullptr) {      // the heads' epilogues have staged d(loss)/d(out) and summed the loss
        HIPCHK(launch_loss_finalize(E.zero + P.loss_acc, h->fused_loss_out, s));
        h->loss_done = 1;
    }
    h->fused_loss_out = nullptr;
    h->last_x = x;
    return CUNET_OK;
}

int cunet_forward_bf16(cunet_plan_t* h, const float* x, float* const* heat, int training, void* stream) {
    if (!h || !x) return fail(CUNET_ERR_INVALID, "null argument");
    if (!h->ws) return fail(CUNET_ERR_STATE, "cunet_bind has not been called");
    Plan& P = h->plan;
    if (training && !h->bound_training) return fail(CUNET_ERR_STATE, "plan is not bound for training");
    if (h->qin_bits) return fail(CUNET_ERR_STATE, "the quantised-input mode (cunet_set_quant_input) is fp32 only");
    const int64_t off16 = h->bound_training ? P.off_bf16_train : P.off_bf16;
    if (h->ws_bytes < off16 + P.n_floats_infer * 2) return fail(CUNET_ERR_STATE, "workspace has no bf16 arena: size it with cunet_workspace_bytes(plan, 2 or 3)");
    hipStream_t s = (hipStream_t)stream;
    h->stem_fused_now = false;
    Exec E(h);
    const int cus = h->num_cus;
    unsigned short* a16 = reinterpret_cast<unsigned short*>(h->ws + off16);      // bf16 arena, element offsets as the fp32 layout
    HIPCHK(hipMemsetAsync(h->ws + P.off_zero, 0, (size_t)P.zero_bytes, s));
    HIPCHK(launch_repack(reinterpret_cast<const RepackEntry*>(h->ws + P.off_repack_tab), (int)P.convs.size(), h->params, E.wsf, s));
    HIPCHK(launch_repack_bf16(reinterpret_cast<const RepackEntry*>(h->ws + P.off_repack_tab), (int)P.convs.size(), h->params, a16,
                              training == 2, s));
    const bool fork_heads = training && h->use_side && h->side && !h->done_ev.empty() && P.opts.heads_on_side;
    const hipStream_t s_main = s;
    std::vector<int> forked_heads;
    for (size_t ni = 0; ni < P.nodes.size(); ++ni) {
        const Node& n = P.nodes[ni];
        const TensorInfo& o = P.tensors[n.out];
        if (n.type == N_STEM_CONV) {               // fp32 kernels on the fp32 image
            const ConvInfo& c = P.convs[n.conv];
            ConvArgs a{};
            a.nseg = 0; a.Ccat = 0; a.training = training ? 1 : 0;
            a.K = c.Cin; a.taps = 1; a.wB = E.wsf + c.wF; a.Kpad = c.KpadF; a.Npad = c.NpadF;
            a.y = E.act(n.out); a.ldy = o.ld; a.Nout = c.Cout; a.ystats = training ? E.stats(n.out) : nullptr;
            a.M = (int)o.rows(); a.H = o.H; a.W = o.W;
            a.img = x; a.IH = P.cfg.height; a.IW = P.cfg.width;
            a.split = (P.opts.f32_split && P.opts.stem_split) ? 1 : 0;
            HIPCHK(launch_conv(a, LD_STEM, EP_FWD, cus, s));
        } else if (n.type == N_STEM_BNPOOL) {
            const int tin = n.segs[0].tensor;
            const TensorInfo& ti = P.tensors[tin];
            const BnInfo& b = P.bns[n.bn];
            PoolArgs a{};
            a.x = E.act(tin); a.y = E.act(n.out); a.ystats = nullptr;       // statistics come from the bf16 copy below
            a.N = ti.N; a.H = ti.H; a.W = ti.W; a.C = ti.C; a.training = training ? 1 : 0;
            a.xstats = E.stats(tin); a.count = (double)ti.rows();
            a.gamma = h->params + b.gamma; a.beta = h->params + b.beta;
            a.rmean = h->buffers + b.rmean; a.rvar = h->buffers + b.rvar;
            HIPCHK(launch_pool_fwd(a, 1, cus, s));
            if (o.ld != o.C || (o.rows() * o.C) % 8) return fail(CUNET_ERR_INVALID, "bf16 path: channel counts must be multiples of 8");
            HIPCHK(launch_cvt_bf16(E.act(n.out), a16 + o.act, training ? E.stats(n.out) : nullptr, (long)o.rows(), o.C, cus, s));
        } else if (n.type == N_POOL) {
            const int tin = n.segs[0].tensor;
            const TensorInfo& ti = P.tensors[tin];
            if (ti.C % 8 || ti.ld != ti.C) return fail(CUNET_ERR_INVALID, "bf16 path: channel counts must be multiples of 8");
            hipError_t e = launch_pool_bf16(a16 + ti.act, a16 + o.act, training ? E.stats(n.out) : nullptr, ti.N, ti.H, ti.W, ti.C, cus, s);
            if (e == hipErrorInvalidValue) return fail(CUNET_ERR_INVALID, "bf16 path: unsupported channel count in a pooled tensor");
            HIPCHK(e);
        } else {  // N_CONV
            const ConvInfo& c = P.convs[n.conv];
            if (n.pair && ni + 1 < P.nodes.size()) {      // ahead + skip adapter of a down block in one launch (see cunet_forward)
                const ConvArgs a0 = conv_fwd_args_bf16(h, E, n, a16, training), a1 = conv_fwd_args_bf16(h, E, P.nodes[ni + 1], a16, training);
                int slotp;
                HIPCHK(prof_begin(h, PC_C1F16, s, slotp));
                const hipError_t ep = launch_conv_bf16_pair(a0, a1, cus, s);
                if (ep == hipSuccess) {
                    HIPCHK(prof_end(h, slotp, 2.0 * 2.0 * a0.M * a0.K * a0.Nout, 2.0 * 2.0 * (double)a0.M * (a0.K + a0.Nout), s));
                    ++ni;
                    continue;
                }
                prof_cancel(h, slotp);
                if (ep != hipErrorNotSupported && ep != hipErrorInvalidValue) HIPCHK(ep);
            }
            ConvArgs a = conv_fwd_args_bf16(h, E, n, a16, training);
            const int is_head = n.head >= 0;
            const bool forked = is_head && fork_heads && ni + 1 < P.nodes.size();      // a training pass: the head runs on the side stream (see cunet_forward; the last node stays)
            if (forked) {
                HIPCHK(hipEventRecord(h->fork_ev[ni], s_main));
                HIPCHK(hipStreamWaitEvent(h->side, h->fork_ev[ni], 0));
                s = h->side;
            }
            if (is_head && training && h->fused_loss_out != nullptr) {
                set_fused_mse(h, E, n, a);
                a.mse_gbf16 = training == 2;             // bf16 gradient tensors
                a.mse_ldd = head_grad_ld(P, n, training == 2);
            }
            int slot_;
            HIPCHK(prof_begin(h, c.taps == 9 ? PC_C3F16 : PC_C1F16, s, slot_));
            hipError_t e = launch_conv_bf16(a, is_head, cus, s);
            if (e == hipErrorInvalidValue)
                return fail(CUNET_ERR_INVALID, "bf16 path: channel counts must be multiples of 32 and rows of 32 (node " + n.name + ")");
            HIPCHK(e);
            HIPCHK(prof_end(h, slot_, 2.0 * a.M * a.K * a.Nout * a.taps, 2.0 * (double)a.M * (a.K + a.Nout), s));
            if (forked) {
                HIPCHK(hipEventRecord(h->done_ev[ni], h->side));
                forked_heads.push_back((int)ni);
                s = s_main;
            }
        }
    }
    for (int ni : forked_heads) HIPCHK(hipStreamWaitEvent(s_main, h->done_ev[ni], 0));      // nothing may be left floating
    if (training)
        HIPCHK(launch_running_update(reinterpret_cast<const RunStatEntry*>(h->ws + P.off_runstat_tab), P.n_runstat,
                                     E.zero, h->buffers, h->counters, 0, s));
    if (heat) {
        for (size_t i = 0; i < P.head_tensors.size(); ++i) {
            if (!heat[i]) continue;
            const TensorInfo& t = P.tensors[P.head_tensors[i]];
            HIPCHK(launch_transpose(E.act(P.head_tensors[i]), heat[i], t.N, t.C, t.H * t.W, t.ld, 0, s));
        }
    }
    h->fwd_training_done = training ? (training == 2 ? 3 : 2) : 0;        // 2: activations are in the bf16 arena; 3: gradient tensors bf16 too
    h->loss_done = 0;
    if (training && h->fused_loss_out != nullptr) {
        HIPCHK(launch_loss_finalize(E.zero + P.loss_acc, h->fused_loss_out, s));
        h->loss_done = 1;
    }
    h->fused_loss_out = nullptr;
    h->last_x = x;
    return CUNET_OK;
}

int cunet_loss_mse(cunet_plan_t* h, const float* target, float* loss, void* stream) {
    if (!h || !target || !loss) return fail(CUNET_ERR_INVALID, "null argument");
    if (!h->ws || !h->bound_training) return fail(CUNET_ERR_STATE, "plan is not bound for training");
    hipStream_t s = (hipStream_t)stream;
    Exec E(h);
    Plan& P = h->plan;
    const TensorInfo& t0 = P.tensors[P.head_tensors[0]];
    float* tgt = E.wsf + P.target_off;
    h->fused_loss_out = nullptr;
    HIPCHK(launch_transpose(target, tgt, t0.N, t0.C, t0.H * t0.W, t0.ld, 1, s));
    double* acc = E.zero + P.loss_acc;
    HIPCHK(hipMemsetAsync(acc, 0, 8, s));
    for (const Node& n : P.nodes) {
        if (n.head < 0) continue;
        const TensorInfo& t = P.tensors[n.out];
        HIPCHK(launch_mse(E.act(n.out), tgt, E.grad(n.out), acc, (long)t.rows(), t.C, t.ld, head_grad_ld(P, n, E.xmode == 2), E.xmode == 2, h->num_cus, s));
    }
    HIPCHK(launch_loss_finalize(acc, loss, s));
    h->loss_done = 1;
    return CUNET_OK;
}

int cunet_loss_mse_fused(cunet_plan_t* h, const float* target, float* loss, void* stream) {
    if (!h || !target || !loss) return fail(CUNET_ERR_INVALID, "null argument");
    if (!h->ws || !h->bound_training) return fail(CUNET_ERR_STATE, "plan is not bound for training");
    hipStream_t s = (hipStream_t)stream;
    Exec E(h);
    Plan& P = h->plan;
    const TensorInfo& t0 = P.tensors[P.head_tensors[0]];
    HIPCHK(launch_transpose(target, E.wsf + P.target_off, t0.N, t0.C, t0.H * t0.W, t0.ld, 1, s));
    h->fused_loss_out = loss;
    return CUNET_OK;
}

int cunet_num_buckets(const cunet_plan_t* p) { return p ? (int)p->plan.bucket_begin.size() : 0; }

int cunet_bucket_range(const cunet_plan_t* p, int bucket, int64_t* begin, int64_t* count) {
    if (!p || !begin || !count || bucket < 0 || bucket >= (int)p->plan.bucket_begin.size())
        return fail(CUNET_ERR_INVALID, "bad bucket index");
    *begin = p->plan.bucket_begin[bucket];
    *count = p->plan.bucket_count[bucket];
    return CUNET_OK;
}

int cunet_bucket_order(const cunet_plan_t* p, int32_t* order, int capacity) {
    if (!p || !order) return fail(CUNET_ERR_INVALID, "null argument");
    const Plan& P = p->plan;
    int n = 0, cur = P.nodes.empty() ? -1 : P.nodes.back().bucket;
    for (int k = (int)P.nodes.size() - 1; k >= 0; --k) {
        if (P.nodes[k].bucket != cur) { if (n < capacity) order[n] = cur; ++n; cur = P.nodes[k].bucket; }
    }
    if (cur >= 0) { if (n < capacity) order[n] = cur; ++n; }
    return n;
}

int cunet_side_stream_join(cunet_plan_t* h, void* stream) {
    if (!h) return fail(CUNET_ERR_INVALID, "null argument");
    if (h->use_side && h->side) {
        HIPCHK(hipEventRecord(h->join_ev, h->side));
        HIPCHK(hipStreamWaitEvent((hipStream_t)stream, h->join_ev, 0));
    }
    return CUNET_OK;
}

int cunet_backward(cunet_plan_t* h, const float* const* grad_heat, void* stream) {
    return cunet_backward_ex(h, grad_heat, stream, nullptr, nullptr);
}

int cunet_backward_ex(cunet_plan_t* h, const float* const* grad_heat, void* stream, cunet_bucket_cb on_bucket,
                      void* user) {
    if (!h) return fail(CUNET_ERR_INVALID, "null argument");
    if (!h->ws || !h->bound_training) return fail(CUNET_ERR_STATE, "plan is not bound for training");
    if (!h->fwd_training_done) return fail(CUNET_ERR_STATE, "cunet_backward needs a preceding training-mode cunet_forward");
    if (!grad_heat && !h->loss_done) return fail(CUNET_ERR_STATE, "no staged loss gradient: call cunet_loss_mse or pass grad_heat");
    if (grad_heat && h->fwd_training_done == 3) return fail(CUNET_ERR_STATE, "bf16 gradient storage takes its loss gradient from cunet_loss_mse only");
    hipStream_t s = (hipStream_t)stream;
    Exec E(h);
    Plan& P = h->plan;
    if (grad_heat) {
        for (size_t i = 0; i < P.head_tensors.size(); ++i) {
            const TensorInfo& t = P.tensors[P.head_tensors[i]];
            if (!grad_heat[i]) {
                HIPCHK(hipMemsetAsync(E.grad(P.head_tensors[i]), 0, (size_t)t.rows() * t.ld * 4, s));
            } else {
                HIPCHK(launch_transpose(grad_heat[i], E.grad(P.head_tensors[i]), t.N, t.C, t.H * t.W, t.ld, 1, s));
            }
        }
    }
    HIPCHK(hipMemsetAsync(h->grads, 0, (size_t)P.n_params * 4, s));
    h->stem_fused_now = P.opts.stem_fuse_dz && E.xmode != 2 && P.nodes.size() >= 2 && P.nodes[0].type == N_STEM_CONV &&
                        P.nodes[1].type == N_STEM_BNPOOL && wg3_active(P, P.nodes[0], E.xmode) && (P.tensors[P.nodes[0].out].H & 1) == 0;
    // (every error / abort return below leaves the pass incomplete: cunet_debug_materialise must not run on reductions that were never
    // finished -- the guard clears the flag unless the pass reaches its end)
    struct FusedGuard { cunet_plan* p; bool keep = false; ~FusedGuard() { if (!keep) p->stem_fused_now = false; } } fused_guard{h};
    // The heads' backward depends on nothing but the loss gradient: all of it (data and weight gradient) goes to the side stream
    // up front, last U-Net first (the order the caller's stream will want the results in), one hand-over for all of them; the
    // caller's stream waits for head k's completion where head k's turn would have been.
    // (round 4: a head's done event is recorded behind its DATA gradient -- what the caller's stream waits for -- and the heads' weight
    // gradients follow behind all of them; heads_on_side = 2: the last U-Net's head, whose data gradient the caller's stream needs before
    // anything else, runs that data gradient on the caller's stream itself)
    std::vector<char> head_done(P.nodes.size(), 0);
    if (h->use_side && h->side && P.opts.heads_on_side) {
        int first = -1;
        std::vector<int> heads;
        for (int k = (int)P.nodes.size() - 1; k >= 0; --k) {
            const Node& n = P.nodes[k];
            if (n.type != N_CONV || n.head < 0) continue;
            const bool own = first < 0 && P.opts.heads_on_side >= 2;
            if (own) {
                const int rc = bwd_node(h, n, k, s, s, BWD_MAIN);
                if (rc != CUNET_OK) return rc;
            }
            if (first < 0) {
                first = k;
                HIPCHK(hipEventRecord(h->fork_ev[k], s));
                HIPCHK(hipStreamWaitEvent(h->side, h->fork_ev[k], 0));
            }
            if (!own) {
                const int rc = bwd_node(h, n, k, h->side, h->side, BWD_MAIN);
                if (rc != CUNET_OK) return rc;
                HIPCHK(hipEventRecord(h->done_ev[k], h->side));
            }
            head_done[k] = own ? 2 : 1;
            heads.push_back(k);
        }
        for (int k : heads) {
            const int rc = bwd_node(h, P.nodes[k], k, h->side, h->side, BWD_WGRAD);
            if (rc != CUNET_OK) return rc;
        }
    }
    int cur_bucket = P.nodes.empty() ? -1 : P.nodes.back().bucket;
    int bucket_hi = (int)P.nodes.size();                       // nodes [k+1, bucket_hi) belong to cur_bucket
    std::vector<int> pending;                                  // nodes whose weight gradient has not been forked yet
    // Fused data + weight gradients write their partial tiles from the CALLER's stream; the bucket's reduce runs on the side stream.
    // (1) the reduce has to see every data gradient of its bucket: when no weight-gradient hand-over follows the bucket's last data
    // gradient, one event does; (2) the partial region alternates between two halves by bucket position, and the caller's stream
    // waits for the reduce of position p - 2 before position p's first fused launch.
    const bool side_on = h->use_side && h->side;
    bool any_fused = false;
    for (size_t k = 0; k < P.nodes.size() && !any_fused; ++k) any_fused = wgrad_is_fused(h, (int)k, E.xmode);
    int position = 0;
    // round 6 (planner option stem_wgrad_caller): the stem's weight gradient -- the last long kernel of a step, 160 us with nothing left to
    // run beside it -- is launched on the CALLER's stream behind the stem's BatchNorm pass instead of queueing behind the last bucket's
    // weight gradients and reduce on the side stream (which are still running then), and its reduce follows it there.
    const bool stem_on_caller = side_on && P.opts.stem_wgrad_caller && !P.nodes.empty() && P.nodes[0].type == N_STEM_CONV && P.nodes[0].wg3_S > 0;
    auto close_bucket = [&](int k_lo, int k_hi) -> int {     // everything that writes bucket `cur_bucket` (nodes [k_lo, k_hi)) has been enqueued
        const bool had_pending = !pending.empty();
        const int rcf = fork_wgrads(h, pending, s);
        if (rcf != CUNET_OK) return rcf;
        if (side_on && any_fused && !had_pending) {
            HIPCHK(hipEventRecord(h->bucket_ev, s));
            HIPCHK(hipStreamWaitEvent(h->side, h->bucket_ev, 0));
        }
        // (the stem's bucket with stem_on_caller: its one weight gradient ran on the caller's stream -- so does its reduce)
        const bool red_on_side = side_on && !(stem_on_caller && cur_bucket == P.cfg.layer_num);
        const int rcr = reduce_wgrad3(h, P.wgred_first[cur_bucket], P.wgred_count[cur_bucket], P.wgred_maxnumel[cur_bucket], red_on_side ? h->side : s);
        if (rcr != CUNET_OK) return rcr;
        if (red_on_side && (any_fused || stem_on_caller)) HIPCHK(hipEventRecord(h->red_ev[position & 1], h->side));
        ++position;
        if (side_on && any_fused && position >= 2) HIPCHK(hipStreamWaitEvent(s, h->red_ev[position & 1], 0));      // (recorded at position - 2)
        return bn_param_grads(h, k_lo, k_hi, cur_bucket, s);      // (on the side stream instead: -0.5 % on the CU-Net-2 step, round 6 -- the side stream is the critical path at the end of a step)
    };
    // (bf16 gradient tensors: shorter kernels, the hand-over bubble weighs more -- 8 per group measured best there, 4 in fp32)
    // (fp32 gradients, round 5: 0 = by depth -- 2 for up to four U-Nets, where starting the side stream's work sooner shortens the tail of the
    // step by more than the extra hand-overs cost (CU-Net-2: 4243-4298 img/s at 2 against 4188-4206 at 4, 4177-4199 at 1, 4190-4198 at 3 on
    // one box), 4 for deeper networks, where the tail is a small part of the step and the hand-overs add up (CU-Net-16: 578 vs 581))
    const int group_f32 = P.opts.wgrad_fork_group > 0 ? P.opts.wgrad_fork_group : (P.cfg.layer_num <= 4 ? 2 : 4);
    const size_t group = (h->use_side && h->side) ? (size_t)std::max(1, E.xmode == 2 ? P.opts.wgrad_fork_group_bf16 : group_f32) : 1;
    int gathered_early = -1;                                   // node whose output gradient the fused pool launch has gathered already
    for (int k = (int)P.nodes.size() - 1; k >= 0; --k) {
        const Node& n = P.nodes[k];
        if (n.bucket != cur_bucket) {      // everything that writes bucket `cur_bucket` has been enqueued
            const int rcb = close_bucket(k + 1, bucket_hi);
            if (rcb != CUNET_OK) return rcb;
            if (on_bucket && on_bucket(cur_bucket, user) != 0) {   // the consumer joins the side stream itself (cunet_side_stream_join)
                (void)cunet_side_stream_join(h, stream);
                h->fwd_training_done = 0;
                return fail(CUNET_ERR_CALLBACK, "bucket callback failed for bucket " + std::to_string(cur_bucket) + ": backward aborted, gradients incomplete");
            }
            cur_bucket = n.bucket;
            bucket_hi = k + 1;
        }
        if (head_done[k]) {                // a head: its backward is on the side stream already (2: its data gradient ran on this stream)
            if (head_done[k] == 1) HIPCHK(hipStreamWaitEvent(s, h->done_ev[k], 0));
            continue;
        }
        if (n.type == N_POOL) {            // gather + pool backward + the skip adapter's gather in one launch where the pattern fits
            int rcp = CUNET_OK;
            const int fused = bwd_pool_with_skip_gather(h, k, s, rcp);
            if (fused < 0) return rcp;
            if (fused) { gathered_early = k - 1; continue; }
        }
        // the skip adapter of a pair (Node::pair on the node in front of it): both adapters' gradients are gathered first, then
        // their data gradients share a launch
        const bool paired = k >= 1 && P.nodes[k - 1].pair && P.nodes[k - 1].bucket == n.bucket;
        // round 6: the gather of a single-consumer output folded into this node's data gradient (fz_eligible): no gather launch, the data
        // gradient writes the tensor as it goes, and the node's weight gradient -- which reads it -- is handed over behind that launch
        const bool fz_try = !paired && k != gathered_early && fz_eligible(h, k, E.xmode);
        for (int q = k; q >= (paired ? k - 1 : k); --q) {
            const Node& nq = P.nodes[q];
            if (fz_try) break;
            if (P.tensors[nq.out].ccount > 0 && q != gathered_early) {  // d(loss)/d(out): gather from the consumers (heads get theirs from the loss)
                const int rcg = gather_tensor_grad(h, nq.out, -1, s);
                if (rcg != CUNET_OK) return rcg;
            }
            if (stem_on_caller && q == 0 && nq.type == N_STEM_CONV) {
                const int rcf = fork_wgrads(h, pending, s);                     // (whatever is still waiting goes to the side stream first)
                if (rcf != CUNET_OK) return rcf;
                // its partial tiles go to the region the bucket two positions back used: that bucket's reduce (side stream) must be through
                if (position >= 2) HIPCHK(hipStreamWaitEvent(s, h->red_ev[position & 1], 0));
                const int rcw = bwd_node(h, nq, 0, s, s, BWD_WGRAD);
                if (rcw != CUNET_OK) return rcw;
                continue;
            }
            if (node_has_wgrad(nq) && !wgrad_is_fused(h, q, E.xmode)) {          // (its d(loss)/d(out) is enqueued: the weight gradient may start once that has run)
                pending.push_back(q);
                if ((pending.size() >= group && q == (paired ? k - 1 : k)) || q == 0) {
                    const int rcf = fork_wgrads(h, pending, s);
                    if (rcf != CUNET_OK) return rcf;
                }
            }
        }
        int launched = 0;
        if (paired) {
            int rcp = CUNET_OK;
            launched = bwd_dgrad_pair(h, k - 1, k, s, rcp);
            if (launched < 0) return rcp;
        }
        if (fz_try) {
            h->fz_node = k;
            int rc = bwd_node(h, n, k, s, s, BWD_MAIN);
            h->fz_node = -1;
            if (rc == CUNET_FZ_FALLBACK) {                       // no fused kernel for this shape: the two launches
                const int rcg = gather_tensor_grad(h, n.out, -1, s);
                if (rcg != CUNET_OK) return rcg;
                rc = bwd_node(h, n, k, s, s, BWD_MAIN);
            }
            if (rc != CUNET_OK) return rc;
            launched = 1;
            if (!wgrad_is_fused(h, k, E.xmode)) {
                pending.push_back(k);
                if (pending.size() >= group || k == 0) {
                    const int rcf = fork_wgrads(h, pending, s);
                    if (rcf != CUNET_OK) return rcf;
                }
            }
        }
        for (int q = k; !launched && q >= (paired ? k - 1 : k); --q) {
            const int rc = bwd_node(h, P.nodes[q], q, s, s, BWD_MAIN);
            if (rc != CUNET_OK) return rc;
        }
        if (paired) --k;
    }
    if (cur_bucket >= 0) {
        const int rcb = close_bucket(0, bucket_hi);
        if (rcb != CUNET_OK) return rcb;
    } else {
        const int rcf = fork_wgrads(h, pending, s);
        if (rcf != CUNET_OK) return rcf;
    }
    if (h->use_side && h->side) {
        HIPCHK(hipEventRecord(h->join_ev, h->side));
        HIPCHK(hipStreamWaitEvent(s, h->join_ev, 0));
    }
    if (on_bucket && cur_bucket >= 0 && on_bucket(cur_bucket, user) != 0) {
        h->fwd_training_done = 0;
        return fail(CUNET_ERR_CALLBACK, "bucket callback failed for bucket " + std::to_string(cur_bucket) + ": gradients not reduced");
    }
    // the reference re-runs every checkpointed cat->BN->ReLU->conv during backward, which updates
    // those BNs' running statistics a second time (models/cu_net.py:30-31,58-59)
    HIPCHK(launch_running_update(reinterpret_cast<const RunStatEntry*>(h->ws + P.off_runstat_tab), P.n_runstat,
                                 E.zero, h->buffers, h->counters, 1, s));
    h->fwd_training_done = 0;
    fused_guard.keep = true;
    return CUNET_OK;
}

int cunet_rmsprop_step(float* params, const float* grads, float* square_avg, int64_t n, double lr, double alpha,
                       double eps, double grad_scale, void* stream) {
    if (!params || !grads || !square_avg || n < 0) return fail(CUNET_ERR_INVALID, "null argument");
    if (((uintptr_t)params & 15) || ((uintptr_t)grads & 15) || ((uintptr_t)square_avg & 15))
        return fail(CUNET_ERR_INVALID, "arenas must be 16-byte aligned");
    HIPCHK(launch_rmsprop(params, grads, square_avg, (long)n, (float)lr, (float)alpha, (float)(1.0 - alpha), (float)eps, (float)grad_scale, (hipStream_t)stream));
    return CUNET_OK;
}

int cunet_get_preds(const float* heat, float* preds, int n, int k, int hh, int w, void* stream) {
    if (!heat || !preds || n < 1 || k < 1 || hh < 1 || w < 1) return fail(CUNET_ERR_INVALID, "bad argument");
    HIPCHK(launch_get_preds(heat, preds, n * k, hh, w, (hipStream_t)stream));
    return CUNET_OK;
}

int cunet_final_preds(const float* heat, const float* center, const float* scale, float* preds, int n, int k, int hh,
                      int w, int res0, int res1, void* stream) {
    if (!heat || !center || !scale || !preds || n < 1 || k < 1 || hh < 1 || w < 1) return fail(CUNET_ERR_INVALID, "bad argument");
    // the refinement reads hm[py-1][px] / hm[py-1][px-2] ... for 1 < px < res0, 1 < py < res1 (pylib/Evaluation.py:113-119):
    // a `res` beyond the map, or a non-square map (where the reference's y = floor(idx / H) can exceed H), would index
    // out of bounds -- the reference raises IndexError there
    if (res0 < 1 || res1 < 1 || res0 > w || res1 > hh) return fail(CUNET_ERR_INVALID, "final_preds: res exceeds the heat map");
    if (hh != w) return fail(CUNET_ERR_INVALID, "final_preds: square heat maps only (the reference's y = floor(idx / size(2)) + 1 is a row index only then)");
    HIPCHK(launch_final_preds(heat, center, scale, nullptr, preds, n, k, hh, w, res0, res1, (hipStream_t)stream));
    return CUNET_OK;
}

int cunet_final_preds_affine(const float* heat, const double* inv, float* preds, int n, int k, int hh, int w, int res0, int res1, void* stream) {
    if (!heat || !inv || !preds || n < 1 || k < 1 || hh < 1 || w < 1) return fail(CUNET_ERR_INVALID, "bad argument");
    if (res0 < 1 || res1 < 1 || res0 > w || res1 > hh) return fail(CUNET_ERR_INVALID, "final_preds: res exceeds the heat map");
    if (hh != w) return fail(CUNET_ERR_INVALID, "final_preds: square heat maps only (the reference's y = floor(idx / size(2)) + 1 is a row index only then)");
    HIPCHK(launch_final_preds(heat, nullptr, nullptr, inv, preds, n, k, hh, w, res0, res1, (hipStream_t)stream));
    return CUNET_OK;
}

int cunet_render_targets(const double* pts, const float* patch, int half, float* out, int nk, int hh, int w, void* stream) {
    if (!pts || !patch || !out || half < 0 || nk < 1 || hh < 1 || w < 1) return fail(CUNET_ERR_INVALID, "bad argument");
    HIPCHK(launch_render_targets(pts, patch, half, out, nk, hh, w, (hipStream_t)stream));
    return CUNET_OK;
}

int cunet_augment_batch(const void* table_dev, const void* table_host, int n, float* out, int res, void* stream) {
    if (!table_dev || !table_host || !out || n < 1 || res < 1) return fail(CUNET_ERR_INVALID, "bad argument");
    const hipError_t e = launch_augment(reinterpret_cast<const AugSample*>(table_dev), reinterpret_cast<const AugSample*>(table_host), n, out, res,
                                        (hipStream_t)stream);
    if (e == hipErrorInvalidValue) return fail(CUNET_ERR_INVALID, "cunet_augment_batch: inconsistent sample record (sizes / scratch pointers)");
    HIPCHK(e);
    return CUNET_OK;
}

int cunet_flip_merge(const float* a, const float* b, const int32_t* perm, float* out, int n, int k, int hh, int w,
                     void* stream) {
    if (!a || !b || !perm || !out || n < 1 || k < 1 || hh < 1 || w < 1) return fail(CUNET_ERR_INVALID, "bad argument");
    HIPCHK(launch_flip_merge(a, b, perm, out, n, k, hh, w, (hipStream_t)stream));
    return CUNET_OK;
}

static int device_cus() {
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 256;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) return 256;
    return cus;
}

int cunet_quant_prepare(float* params, float* saved, const void* table, int nconv, int max_o, int max_n, int bits_w,
                        int bits_g, int keep_scale, void* stream) {
    if (!params || !saved || !table || nconv < 1 || max_o < 1 || max_n < 1) return fail(CUNET_ERR_INVALID, "bad argument");
    if (bits_w < 1 || bits_g < 1 || (size_t)max_n * 4 > 60 * 1024) return fail(CUNET_ERR_INVALID, "unsupported bit width or filter size");
    HIPCHK(launch_quant_prepare((const QuantEntry*)table, nconv, max_o, max_n, params, saved, bits_w, bits_g, keep_scale, (hipStream_t)stream));
    return CUNET_OK;
}

int cunet_quant_restore(float* params, const float* saved, const void* table, int nconv, void* stream) {
    if (!params || !saved || !table || nconv < 1) return fail(CUNET_ERR_INVALID, "bad argument");
    HIPCHK(launch_quant_restore((const QuantEntry*)table, nconv, params, saved, (hipStream_t)stream));
    return CUNET_OK;
}

int cunet_quant_grad(const float* params, float* grads, const void* table, int nconv, int max_o, int bits_w, int bits_g,
                     int keep_scale, void* stream) {
    if (!params || !grads || !table || nconv < 1 || max_o < 1) return fail(CUNET_ERR_INVALID, "bad argument");
    HIPCHK(launch_quant_grad((const QuantEntry*)table, nconv, max_o, params, grads, bits_w, bits_g, keep_scale, (hipStream_t)stream));
    return CUNET_OK;
}

int cunet_ternary_pack(const float* w, uint64_t* wpos, uint64_t* wneg, int o, int c, int taps, void* stream) {
    if (!w || !wpos || !wneg || o < 1 || c < 1 || (taps != 1 && taps != 9)) return fail(CUNET_ERR_INVALID, "bad argument");
    HIPCHK(launch_ternary_pack(w, wpos, wneg, o, c, taps, round_up(o, 64), (hipStream_t)stream));
    return CUNET_OK;
}

int cunet_ternary_conv(const float* x, const float* scale, const float* shift, const uint64_t* wpos, const uint64_t* wneg,
                       float* y, int n, int hh, int w, int c, int o, int taps, int bits_i, void* stream) {
    if (!x || !scale || !shift || !wpos || !wneg || !y || n < 1 || hh < 1 || w < 1) return fail(CUNET_ERR_INVALID, "bad argument");
    TernArgs a{};
    a.x = x; a.scale = scale; a.shift = shift; a.wpos = wpos; a.wneg = wneg; a.y = y;
    a.M = n * hh * w; a.H = hh; a.W = w; a.C = c; a.O = o; a.Opad = round_up(o, 64); a.taps = taps; a.bits_i = bits_i;
    a.ldx = c; a.ldy = o;
    hipError_t e = launch_ternary_conv(a, device_cus(), (hipStream_t)stream);
    if (e != hipSuccess) return fail(e == hipErrorInvalidValue ? CUNET_ERR_INVALID : CUNET_ERR_HIP, std::string("ternary conv: ") + hipGetErrorString(e));
    return CUNET_OK;
}

int cunet_ternary_conv_ex(const float* x, const float* scale, const float* shift, const uint64_t* wpos, const uint64_t* wneg,
                          uint64_t* planes, float* y, double* ystats, int n, int hh, int w, int c, int o, int taps, int bits_i,
                          int variant, void* stream) {
    if (!x || !scale || !shift || !wpos || !wneg || !planes || !y || n < 1 || hh < 1 || w < 1) return fail(CUNET_ERR_INVALID, "bad argument");
    if (c > 128 || bits_i > 8 || bits_i < 2) return fail(CUNET_ERR_INVALID, "ternary conv on bit-plane records: C <= 128, 2 <= bits_i <= 8");
    if (o < 1 || c < 1 || (taps != 1 && taps != 9)) return fail(CUNET_ERR_INVALID, "ternary conv: O >= 1, C >= 1, taps 1 or 9");
    if ((uintptr_t)y & 15) return fail(CUNET_ERR_INVALID, "ternary conv: y must be 16-byte aligned (the pixel kernel stores 16-byte pieces)");
    TernArgs a{};
    a.x = x; a.scale = scale; a.shift = shift; a.wpos = wpos; a.wneg = wneg; a.y = y;
    a.M = n * hh * w; a.H = hh; a.W = w; a.C = c; a.O = o; a.Opad = round_up(o, 64); a.taps = taps; a.bits_i = bits_i;
    a.ldx = c; a.ldy = o;
    a.planes = planes; a.ystats = ystats; a.variant = variant;
    hipError_t e = launch_ternary_conv(a, device_cus(), (hipStream_t)stream);
    if (e != hipSuccess) return fail(e == hipErrorInvalidValue ? CUNET_ERR_INVALID : CUNET_ERR_HIP, std::string("ternary conv: ") + hipGetErrorString(e));
    return CUNET_OK;
}

int cunet_debug_materialise(cunet_plan_t* h, void* stream) {
    // planner option stem_fuse_dz: the last backward never wrote d(loss)/d(conv0 output) -- the stem's weight gradient derived it on the
    // fly.  A debugger that wants to LOOK at that tensor gets it written now by the second stem pass, from the same inputs (the first pass's
    // reductions and the pooled features' gradient are still in the workspace until the next forward): the values the fused kernel used.
    if (!h) return fail(CUNET_ERR_INVALID, "null argument");
    if (!h->stem_fused_now || !h->ws || !h->bound_training) return CUNET_OK;      // (a forward since then clears the flag: its inputs are gone)
    hipStream_t s = (hipStream_t)stream;
    Plan& P = h->plan;
    Exec E(h);
    const Node& n = P.nodes[1];
    const int tin = n.segs[0].tensor;
    const TensorInfo& ti = P.tensors[tin];
    const BnInfo& b = P.bns[n.bn];
    PoolArgs a{};
    a.x = E.act(tin); a.gy = E.grad(n.out); a.gx = E.grad(tin);
    a.xstats = E.stats(tin); a.count = (double)ti.rows();
    a.gamma = h->params + b.gamma; a.beta = h->params + b.beta;
    a.N = ti.N; a.H = ti.H; a.W = ti.W; a.C = ti.C; a.training = 1;
    a.red = E.zero + n.red;
    a.xbf16 = 0;
    HIPCHK(launch_stem_bwd(a, 3, nullptr, nullptr, h->num_cus, s));      // (3: the dz pass alone -- dgamma / dbeta are in the arena already)
    h->stem_fused_now = false;      // once: the tensor exists now (a later debug_poke of it must not be overwritten by a second materialise)
    return CUNET_OK;
}

int cunet_debug_run_node_backward(cunet_plan_t* h, int node, void* stream) {
    if (!h || node < 0 || node >= (int)h->plan.nodes.size()) return fail(CUNET_ERR_INVALID, "bad node index");
    if (!h->ws || !h->bound_training) return fail(CUNET_ERR_STATE, "plan is not bound for training");
    hipStream_t s = (hipStream_t)stream;
    Plan& P = h->plan;
    const Node& n = P.nodes[node];
    if (n.red >= 0)
        HIPCHK(hipMemsetAsync(h->ws + P.off_zero + 8 * n.red, 0, (size_t)16 * n.Ccat, s));
    HIPCHK(hipMemsetAsync(h->grads, 0, (size_t)P.n_params * 4, s));
    h->stem_fused_now = false;                                      // (a single node: every tensor it reads and writes is a real tensor)
    const bool fused = wgrad_is_fused(h, node, Exec(h).xmode);      // (its weight gradient comes with the data gradient launched below)
    {
        std::vector<int> one;
        if (node_has_wgrad(n) && !fused) one.push_back(node);
        const int rcf = fork_wgrads(h, one, s);
        if (rcf != CUNET_OK) return rcf;
    }
    // an adapter of a pair (Node::pair): its data gradient runs as cunet_backward runs it, in the pair's launch (the partner's
    // d(loss)/d(out) is whatever the workspace holds; its dz / reductions are not read here)
    const int k0 = n.pair ? node : ((node >= 1 && P.nodes[node - 1].pair) ? node - 1 : -1);
    int launched = 0;
    if (k0 >= 0) {
        const Node& partner = P.nodes[k0 == node ? node + 1 : k0];
        if (partner.red >= 0) HIPCHK(hipMemsetAsync(h->ws + P.off_zero + 8 * partner.red, 0, (size_t)16 * partner.Ccat, s));
        int rcp = CUNET_OK;
        launched = bwd_dgrad_pair(h, k0, k0 + 1, s, rcp);
        if (launched < 0) return rcp;
    }
    if (!launched) {
        const int rc = bwd_node(h, n, node, s, s, BWD_MAIN);
        if (rc != CUNET_OK) return rc;
    }
    if (n.wg3_S > 0) {
        const int rcr = reduce_wgrad3(h, n.wg3_entry, 1, (int)P.wg3_numel(n), (h->use_side && h->side && !fused) ? h->side : s);
        if (rcr != CUNET_OK) return rcr;
    }
    if (n.type == N_CONV) {                // this node's contribution to each of its inputs, and its BN parameter gradients
        for (size_t j = 0; j < n.segs.size(); ++j) {
            bool seen = false;
            for (size_t i = 0; i < j; ++i) seen |= (n.segs[i].tensor == n.segs[j].tensor);
            if (seen) continue;
            const int rcg = gather_tensor_grad(h, n.segs[j].tensor, node, s);
            if (rcg != CUNET_OK) return rcg;
        }
        const int rcg = bn_param_grads(h, node, node + 1, -1, s);
        if (rcg != CUNET_OK) return rcg;
    }
    if (h->use_side && h->side) {
        HIPCHK(hipEventRecord(h->join_ev, h->side));
        HIPCHK(hipStreamWaitEvent(s, h->join_ev, 0));
    }
    return CUNET_OK;
}

int cunet_profile_begin(cunet_plan_t* h, int mode, int cls) {
    if (!h || mode < 0 || mode > 2 || (mode == 2 && (cls < 0 || cls >= CUNET_PROF_NCLS))) return fail(CUNET_ERR_INVALID, "bad profile mode");
    h->prof_mode = mode;
    h->prof_cls = cls;
    return CUNET_OK;
}

int cunet_profile_reset(cunet_plan_t* h) {
    if (!h) return fail(CUNET_ERR_INVALID, "null argument");
    for (int w = 0; w < 2; ++w)
        for (int i = 0; i < CUNET_PROF_NCLS; ++i) { h->prof_ms[w][i] = h->prof_flops[w][i] = h->prof_bytes[w][i] = 0; h->prof_count[w][i] = 0; }
    return CUNET_OK;
}

int cunet_profile_collect(cunet_plan_t* h) {
    if (!h) return fail(CUNET_ERR_INVALID, "null argument");
    for (auto& r : h->prof_pending) {
        HIPCHK(hipEventSynchronize(r.b));
        float ms = 0.f;
        HIPCHK(hipEventElapsedTime(&ms, r.a, r.b));
        const int w = r.on_side;
        h->prof_ms[w][r.cls] += ms; h->prof_flops[w][r.cls] += r.flops; h->prof_bytes[w][r.cls] += r.bytes; h->prof_count[w][r.cls] += 1;
        h->prof_pool.push_back(r.a); h->prof_pool.push_back(r.b);
    }
    h->prof_pending.clear();
    return CUNET_OK;
}

int cunet_profile_num_classes(void) { return CUNET_PROF_NCLS; }
const char* cunet_profile_class_name(int cls) { return (cls >= 0 && cls < CUNET_PROF_NCLS) ? kProfNames[cls] : ""; }

int cunet_profile_get(const cunet_plan_t* h, int cls, int64_t* count, double* ms, double* flops, double* bytes) {
    return cunet_profile_get_stream(h, cls, -1, count, ms, flops, bytes);
}

int cunet_profile_get_stream(const cunet_plan_t* h, int cls, int which, int64_t* count, double* ms, double* flops, double* bytes) {
    if (!h || cls < 0 || cls >= CUNET_PROF_NCLS || which < -1 || which > 1 || !count || !ms || !flops || !bytes) return fail(CUNET_ERR_INVALID, "bad argument");
    *count = 0; *ms = *flops = *bytes = 0.0;
    for (int w = 0; w < 2; ++w) {
        if (which >= 0 && which != w) continue;
        *count += h->prof_count[w][cls]; *ms += h->prof_ms[w][cls]; *flops += h->prof_flops[w][cls]; *bytes += h->prof_bytes[w][cls];
    }
    return CUNET_OK;
}

}  // extern "C"
