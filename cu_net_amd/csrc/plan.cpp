// Plan builder: turns create_cu_net(...) hyper-parameters into (a) the reference's state_dict
// layout on flat arenas and (b) a static list of fused nodes over NHWC tensors.
//
// Wiring follows the reference's forward (file:line in /root/reference):
//   models/cu_net.py:336-360  wrapper loop: stem, intermedia, hourglass, heads at loss anchors
//   models/cu_net.py:166-190  intermedia FIFO of depth `order`
//   models/cu_net.py:252-269  4 x (down block -> maxpool), neck, 4 x (upsample -> up block on [x, skip])
//   models/cu_net.py:115-144  dense block: layer i on [inputs + carried], FIFO update, adapters on [.. + new]
// but there are no python lists at run time: every concat is a list of segment references
// resolved here, once.
#include "plan.h"

#include <cmath>
#include <cstdio>
#include <sstream>

#include "common.h"

namespace cunet {

static const int NUM_BLOCKS = 4;   // models/cu_net.py:232

PlannerOptions& planner_options() { static PlannerOptions o; return o; }

void Plan::add_state(const std::string& name, int kind, std::vector<int64_t> shape, int bucket) {
    StateEntry e;
    e.name = name;
    e.kind = kind;
    e.shape = shape;
    e.bucket = bucket;
    e.numel = 1;
    for (auto d : shape) e.numel *= d;
    e.offset = -1;
    if (kind != 0) {                                  // buffers / counters: state order
        int64_t* ctr = kind == 1 ? &n_buffers : &n_counters;
        if (kind == 1) *ctr = round_up64(*ctr, 4);    // keep every float tensor 16-byte aligned in its arena
        e.offset = *ctr;
        *ctr += e.numel;
    }
    state_index[name] = (int)state.size();
    state.push_back(e);
}

// Parameter (== gradient) arena is laid out BUCKET-major, not in state_dict order: bucket i holds
// every parameter that U-Net index i touches (layers.i / adapters_*.i of all nine blocks,
// linears.i, intermedia.adapters.(i-1)), the last bucket holds the stem.  Backward completes the
// buckets in the order L-1, ..., 0, stem, so each one can be all-reduced over RCCL while the
// remaining backward is still running (the reference's DataParallel reduces everything at the end,
// cu-net.py:59).
void Plan::assign_param_offsets() {
    const int nb = cfg.layer_num + 1;
    bucket_begin.assign(nb, 0);
    bucket_count.assign(nb, 0);
    int64_t off = 0;
    for (int b = 0; b < nb; ++b) {
        off = round_up64(off, 64);
        bucket_begin[b] = off;
        for (auto& e : state) {
            if (e.kind != 0 || e.bucket != b) continue;
            off = round_up64(off, 4);
            e.offset = off;
            off += e.numel;
        }
        off = round_up64(off, 4);
        bucket_count[b] = off - bucket_begin[b];
    }
    n_params = round_up64(off, 4);
}

void Plan::bn_entries(const std::string& p, int c, int bucket) {
    add_state(p + ".weight", 0, {c}, bucket);
    add_state(p + ".bias", 0, {c}, bucket);
    add_state(p + ".running_mean", 1, {c}, bucket);
    add_state(p + ".running_var", 1, {c}, bucket);
    add_state(p + ".num_batches_tracked", 2, {}, bucket);
}

// models/cu_net.py:75-112
void Plan::block_entries(const std::string& prefix, int in_num, bool requires_skip, bool is_up) {
    const int g = cfg.growth_rate, L = cfg.layer_num, K = cfg.order;
    const int bott = cfg.neck_size * g;
    for (int i = 0; i < L; ++i) {
        const int cin = in_num + std::min(i, K) * g;
        const std::string p = prefix + ".layers." + std::to_string(i);
        bn_entries(p + ".norm1", cin, i);
        add_state(p + ".conv1.weight", 0, {bott, cin, 1, 1}, i);
        bn_entries(p + ".norm2", bott, i);
        add_state(p + ".conv2.weight", 0, {g, bott, 3, 3}, i);
    }
    const int adapter_out = is_up ? in_num / 2 : in_num;
    for (int which = 0; which < (requires_skip ? 2 : 1); ++which) {
        const std::string nm = which == 0 ? ".adapters_ahead." : ".adapters_skip.";
        for (int i = 0; i < L; ++i) {
            const int cin = in_num + (std::min(i, K) + 1) * g;
            const std::string p = prefix + nm + std::to_string(i);
            bn_entries(p + ".adapter_norm", cin, i);
            add_state(p + ".adapter_conv.weight", 0, {adapter_out, cin, 1, 1}, i);
        }
    }
}

// reference registration order: features, hg.{down_blocks, up_blocks, neck_block}, linears, intermedia
void Plan::build_state() {
    const int c0 = cfg.init_chan_num;
    const int stem_bucket = cfg.layer_num;
    add_state("features.conv0.weight", 0, {c0, 3, 7, 7}, stem_bucket);
    bn_entries("features.norm0", c0, stem_bucket);
    for (int j = 0; j < NUM_BLOCKS; ++j) block_entries("hg.down_blocks." + std::to_string(j), c0, true, false);
    for (int j = 0; j < NUM_BLOCKS; ++j) block_entries("hg.up_blocks." + std::to_string(j), 2 * c0, false, true);
    block_entries("hg.neck_block", c0, false, false);
    for (int i = 0; i < cfg.layer_num; ++i) {
        const std::string p = "linears." + std::to_string(i);
        bn_entries(p + ".norm", c0, i);
        add_state(p + ".conv.weight", 0, {cfg.class_num, c0, 1, 1}, i);
    }
    for (int i = 0; i < cfg.layer_num - 1; ++i) {   // models/cu_net.py:156-162
        const int cin = i < cfg.order ? c0 + (i + 1) * c0 : c0 + cfg.order * c0;
        const std::string p = "intermedia.adapters." + std::to_string(i);
        bn_entries(p + ".adapter_norm", cin, i + 1);
        add_state(p + ".adapter_conv.weight", 0, {c0, cin, 1, 1}, i + 1);
    }
    assign_param_offsets();
    n_buffers = round_up64(n_buffers, 4);
}

int64_t Plan::param(const std::string& n) const {
    auto it = state_index.find(n);
    if (it == state_index.end()) return -1;
    return state[it->second].offset;
}

int Plan::tensor_by_name(const std::string& n) const {
    for (size_t i = 0; i < tensors.size(); ++i)
        if (tensors[i].name == n) return (int)i;
    return -1;
}

int Plan::add_tensor(const std::string& name, int N, int H, int W, int C, bool stats) {
    TensorInfo t;
    t.name = name;
    t.N = N; t.H = H; t.W = W; t.C = C;
    t.ld = round_up(C, 4);
    if (stats) {
        t.stats = n_zero_doubles;
        n_zero_doubles += 2 * (int64_t)C;
    }
    tensors.push_back(t);
    return (int)tensors.size() - 1;
}

int Plan::add_bn(const std::string& path, bool ckpt) {
    BnInfo b;
    b.name = path;
    b.gamma = param(path + ".weight");
    b.beta = param(path + ".bias");
    b.rmean = param(path + ".running_mean");
    b.rvar = param(path + ".running_var");
    b.counter = param(path + ".num_batches_tracked");
    b.C = (int)state[state_index.at(path + ".weight")].numel;
    b.ckpt = ckpt;
    bns.push_back(b);
    return (int)bns.size() - 1;
}

int Plan::add_conv(const std::string& path, int taps, bool need_bwd) {
    const StateEntry& e = state[state_index.at(path + ".weight")];
    ConvInfo c;
    c.name = path;
    c.Cout = (int)e.shape[0];
    c.Cin = (int)e.shape[1] * (taps == 49 ? 49 : 1);   // stem: im2col K = 3*7*7, handled as a 1-tap GEMM
    c.taps = taps == 49 ? 1 : taps;
    c.w = e.offset;
    c.KpadF = round_up(c.Cin, 32);
    c.NpadF = round_up(c.Cout, 32);
    c.KpadB = round_up(c.Cout, 32);
    c.NpadB = round_up(c.Cin, 32);
    c.wB = need_bwd ? 0 : -1;
    convs.push_back(c);
    return (int)convs.size() - 1;
}

int Plan::conv_node(const std::string& name, const std::string& bn_path, const std::string& conv_path,
                    const std::vector<SegRef>& segs, int taps, bool ckpt, int H, int W, int head, bool out_stats) {
    Node n;
    n.type = N_CONV;
    n.name = name;
    n.bn = add_bn(bn_path, ckpt);
    n.conv = add_conv(conv_path, taps, true);
    n.segs = segs;
    n.taps = taps;
    n.head = head;
    int cc = 0;
    for (auto& s : segs) cc += tensors[s.tensor].C;
    n.Ccat = cc;
    if (cc != convs[n.conv].Cin || cc != bns[n.bn].C) {
        error = "internal: channel mismatch at " + name;
        return -1;
    }
    if ((int)segs.size() > MAXSEG) {
        error = "order too large: a concat has more than " + std::to_string(MAXSEG) + " segments at " + name;
        return -1;
    }
    n.red = n_zero_doubles;
    n_zero_doubles += 2 * (int64_t)cc;
    n.bucket = state[state_index.at(conv_path + ".weight")].bucket;
    n.out = add_tensor(conv_path, cfg.batch, H, W, convs[n.conv].Cout, out_stats);
    nodes.push_back(n);
    return n.out;
}

bool Plan::build(const cunet_cfg& c) {
    cfg = c;
    opts = planner_options();
    error.clear();
    // ---- validation (models/cu_net.py:274-287, with exit() turned into an error)
    if (c.layer_num < 1 || c.loss_num < 1 || c.loss_num > c.layer_num) { error = "need 1 <= loss_num <= layer_num"; return false; }
    const double every = (double)c.layer_num / (double)c.loss_num;
    for (int i = 0; i < c.loss_num; ++i) {
        const int a = (int)std::nearbyint(every * (i + 1));   // python round(): half to even
        if (a <= c.layer_num) anchors.push_back(a);
    }
    bool has_last = false;
    for (int a : anchors) has_last |= (a == c.layer_num);
    if (!has_last || (int)anchors.size() != c.loss_num) { error = "loss anchors do not include the last U-Net"; return false; }
    if (c.order >= c.layer_num) { error = "order is larger than the layer number."; return false; }
    if (c.order < 0 || c.neck_size < 1 || c.growth_rate < 1 || c.init_chan_num < 1 || c.class_num < 1) { error = "bad hyper-parameter"; return false; }
    if (c.growth_rate % 4 || c.init_chan_num % 4 || (c.neck_size * c.growth_rate) % 4) {
        error = "this build needs growth_rate, init_chan_num and neck_size*growth_rate to be multiples of 4 (16-byte NHWC pieces)";
        return false;
    }
    if (c.init_chan_num % 2) { error = "init_chan_num must be even"; return false; }
    if (c.batch < 1 || c.height < 64 || c.width < 64 || c.height % 64 || c.width % 64) {
        error = "input must be N>=1 and H, W positive multiples of 64";
        return false;
    }
    if ((int64_t)c.batch * c.height * c.width / 4 > (int64_t)1 << 30) { error = "batch too large for 32-bit row indices"; return false; }

    build_state();

    const int N = c.batch, c0 = c.init_chan_num, g = c.growth_rate, K = c.order, L = c.layer_num;
    const int bott = c.neck_size * g;
    (void)bott;
    const int H4 = c.height / 4, W4 = c.width / 4;

    // ---- stem (models/cu_net.py:299-304)
    const int t_c0 = add_tensor("features.conv0", N, c.height / 2, c.width / 2, c0, true);
    {
        Node n; n.type = N_STEM_CONV; n.name = "features.conv0";
        n.conv = add_conv("features.conv0", 49, false);
        n.out = t_c0;
        n.bucket = cfg.layer_num;
        nodes.push_back(n);
    }
    const int t_x0 = add_tensor("features.pool0", N, H4, W4, c0, true);
    {
        Node n; n.type = N_STEM_BNPOOL; n.name = "features.pool0";
        n.bn = add_bn("features.norm0", false);
        n.segs.push_back({t_c0, 0});
        n.out = t_x0;
        n.Ccat = c0;
        n.red = n_zero_doubles; n_zero_doubles += 2 * (int64_t)c0;
        n.bucket = cfg.layer_num;
        nodes.push_back(n);
    }

    std::map<std::string, std::deque<int>> saved;
    std::deque<int> inter_saved;
    auto dense_block = [&](const std::string& prefix, std::vector<SegRef> inputs, int i, int H, int W,
                           bool requires_skip, int& ahead, int& skip) -> bool {
        std::deque<int>& fifo = saved[prefix];
        std::vector<SegRef> segs = inputs;
        for (int t : fifo) segs.push_back({t, 0});
        const std::string p = prefix + ".layers." + std::to_string(i);
        const int z = conv_node(p + ".conv1", p + ".norm1", p + ".conv1", segs, 1, true, H, W, -1, true);
        if (z < 0) return false;
        const int out = conv_node(p + ".conv2", p + ".norm2", p + ".conv2", {{z, 0}}, 9, false, H, W, -1, true);
        if (out < 0) return false;
        if (i < K) fifo.push_back(out);                      // models/cu_net.py:133-137
        else if (!fifo.empty()) { fifo.pop_front(); fifo.push_back(out); }
        segs.push_back({out, 0});                            // :138
        const std::string pa = prefix + ".adapters_ahead." + std::to_string(i);
        ahead = conv_node(pa + ".adapter_conv", pa + ".adapter_norm", pa + ".adapter_conv", segs, 1, true, H, W, -1, true);
        if (ahead < 0) return false;
        skip = -1;
        if (requires_skip) {
            const std::string ps = prefix + ".adapters_skip." + std::to_string(i);
            skip = conv_node(ps + ".adapter_conv", ps + ".adapter_norm", ps + ".adapter_conv", segs, 1, true, H, W, -1, true);
            if (skip < 0) return false;
            if (opts.pair_adapters) nodes[nodes.size() - 2].pair = 1;      // (conv_node appended the two adapters back to back)
        }
        return true;
    };

    int cur = t_x0;
    int head_count = 0;
    for (int i = 0; i < L; ++i) {
        // intermedia (models/cu_net.py:166-190)
        if (i == 0) {
            inter_saved.clear();
            if (K != 0) inter_saved.push_back(cur);
        } else {
            std::vector<SegRef> segs{{cur, 0}};
            for (int t : inter_saved) segs.push_back({t, 0});
            const std::string p = "intermedia.adapters." + std::to_string(i - 1);
            const int out = conv_node(p + ".adapter_conv", p + ".adapter_norm", p + ".adapter_conv", segs, 1, true, H4, W4, -1, true);
            if (out < 0) return false;
            if (i < K) inter_saved.push_back(out);
            else if (!inter_saved.empty()) { inter_saved.pop_front(); inter_saved.push_back(out); }
            cur = out;
        }
        if (i == 0) saved.clear();
        // hourglass (models/cu_net.py:252-269)
        int h = cur, Hh = H4, Ww = W4;
        int skips[NUM_BLOCKS];
        for (int j = 0; j < NUM_BLOCKS; ++j) {
            int ahead, skip;
            if (!dense_block("hg.down_blocks." + std::to_string(j), {{h, 0}}, i, Hh, Ww, true, ahead, skip)) return false;
            skips[j] = skip;
            Node n; n.type = N_POOL;
            n.name = "hg.down_blocks." + std::to_string(j) + ".pool@" + std::to_string(i);
            n.segs.push_back({ahead, 0});
            Hh /= 2; Ww /= 2;
            n.out = add_tensor(n.name, N, Hh, Ww, tensors[ahead].C, true);
            n.Ccat = tensors[ahead].C;
            n.bucket = i;
            nodes.push_back(n);
            h = n.out;
        }
        {
            int ahead, skip;
            if (!dense_block("hg.neck_block", {{h, 0}}, i, Hh, Ww, false, ahead, skip)) return false;
            h = ahead;
        }
        for (int j = NUM_BLOCKS - 1; j >= 0; --j) {
            Hh *= 2; Ww *= 2;
            int ahead, skip;
            if (!dense_block("hg.up_blocks." + std::to_string(j), {{h, 1}, {skips[j], 0}}, i, Hh, Ww, false, ahead, skip)) return false;
            h = ahead;
        }
        cur = h;
        bool is_anchor = false;
        for (int a : anchors) is_anchor |= (a == i + 1);
        if (is_anchor) {                                       // models/cu_net.py:353-356
            const std::string p = "linears." + std::to_string(i);
            const int ho = conv_node(p + ".conv", p + ".norm", p + ".conv", {{cur, 0}}, 1, false, H4, W4, head_count, false);
            if (ho < 0) return false;
            head_tensors.push_back(ho);
            ++head_count;
        }
    }

    // ---- backward-order check: every node's output gradient exists when backward reaches the node
    {
        std::vector<char> written(tensors.size(), 0);
        for (int h : head_tensors) written[h] = 1;            // d(loss)/d(heat) is provided
        for (int k = (int)nodes.size() - 1; k >= 0; --k) {
            Node& n = nodes[k];
            if (n.type == N_CONV) {
                if (!written[n.out]) { error = "internal: gradient of " + tensors[n.out].name + " never produced"; return false; }
                for (auto& s : n.segs) written[s.tensor] = 1;
            } else if (n.type == N_POOL || n.type == N_STEM_BNPOOL) {
                if (!written[n.out]) { error = "internal: gradient of " + tensors[n.out].name + " never produced"; return false; }
                if (written[n.segs[0].tensor]) { error = "internal: pooled tensor has a second consumer"; return false; }
                written[n.segs[0].tensor] = 1;
            }
        }
    }

    // ---- backward gather lists: the gradient of a tensor is assembled once, from the dz slices of all
    //      conv nodes that read it (a pooled tensor gets its gradient from the pool node alone)
    {
        std::vector<std::vector<Contrib>> by_t(tensors.size());
        for (int k = 0; k < (int)nodes.size(); ++k)
            if (nodes[k].type == N_CONV)
                for (int j = 0; j < (int)nodes[k].segs.size(); ++j) by_t[nodes[k].segs[j].tensor].push_back({k, j});
        contribs.clear();
        for (size_t t = 0; t < tensors.size(); ++t) {
            tensors[t].cfirst = (int)contribs.size();
            tensors[t].ccount = (int)by_t[t].size();
            contribs.insert(contribs.end(), by_t[t].begin(), by_t[t].end());
        }
    }

    // bf16 gradient storage: a head's gradient tensor is padded to the 32-multiple of channels its bf16 MFMA data gradient
    // contracts over when that fits the tensor's fp32 slot (2 * ld bf16 elements), e.g. K = 68 -> 96
    for (auto& t : tensors) t.gld16 = t.ld;
    for (auto& n : nodes)
        if (n.type == N_CONV && n.head >= 0) {
            const int kp = convs[n.conv].KpadB;
            TensorInfo& o = tensors[n.out];
            if (kp % 8 == 0 && 2 * o.ld >= kp) o.gld16 = kp;
        }

    layout_workspace();
    describe();
    return true;
}

void Plan::layout_workspace() {
    int64_t off = 0;
    off_repack_tab = off;
    off += round_up64((int64_t)convs.size() * (int64_t)sizeof(RepackEntry), 256);
    // running-stat table: one entry per (BN, segment)
    n_runstat = 0;
    for (auto& n : nodes) {
        if (n.type == N_CONV) n_runstat += (int)n.segs.size();
        else if (n.type == N_STEM_BNPOOL) n_runstat += 1;
    }
    off_runstat_tab = off;
    off += round_up64((int64_t)n_runstat * (int64_t)sizeof(RunStatEntry), 256);
    // wgrad3 eligibility and split geometry (host-only decision; the kernel's requirements are re-checked at launch).
    // Nodes are visited bucket by bucket so that a bucket's entries are contiguous in the reduce table.
    {
        const int P = 32;
        const PlannerOptions& po = opts;
        const int min_chunks = std::max(1, tune_int("CUNET_WG3_MIN_CHUNKS", po.wgrad3_min_chunks)), smax = std::max(1, tune_int("CUNET_WG3_SMAX", po.wgrad3_max_splits > 0 ? po.wgrad3_max_splits : (po.f32_split ? 192 : 256)));
        const int min_m = tune_int("CUNET_WG3_MIN_M", po.wgrad3_min_rows), enable = tune_int("CUNET_WG3", 1);
        // bf16 gradient tensors (measured on CU-Net-8: 1340 img/s at 256 splits / 2 chunks, 1387 at 96 / 4, 1110 at 32)
        const int min_chunks16 = std::max(1, po.wgrad3_min_chunks_bf16), smax16 = std::max(1, po.wgrad3_max_splits_bf16);
        const int enable3 = tune_int("CUNET_WG3_3X3", 1), min_rows3 = std::max(1, tune_int("CUNET_WG3_3X3_ROWS", 6));     // image rows per workgroup
        const int min_w3 = tune_int("CUNET_WG3_3X3_MIN_W", 2);        // (narrower levels would keep the per-wave kernel)
        const int nb = cfg.layer_num + 1;
        wgred_first.assign(nb, 0); wgred_count.assign(nb, 0); wgred_maxnumel.assign(nb, 0);
        n_wgred = 0;
        for (int b = 0; b < nb; ++b) {
            wgred_first[b] = n_wgred;
            for (auto& n : nodes) {
                if (n.type == N_STEM_CONV && n.bucket == b && enable && po.wgrad3_stem) {
                    // stem 7x7/2: a workgroup owns `rows` output rows of one image (wgrad3_stem_kernel); ~one workgroup per CU
                    const ConvInfo& c = convs[n.conv];
                    const TensorInfo& o = tensors[n.out];
                    const int IW = cfg.width;
                    const bool oks = c.Cout == 128 && o.ld == 128 && c.Cin == STEM_K && o.W % STEM_CHUNK == 0 && IW % 8 == 0 &&
                                     cfg.width == 2 * o.W && cfg.height == 2 * o.H && o.rows() >= min_m;
                    // one output row per workgroup must fit the kernel's LDS ring and its staging registers (wide inputs, IW >= 1664,
                    // do not: they keep the per-wave atomic kernel, wg3_S = 0)
                    const bool fits1 = (int64_t)((((2 * 1 + 6) * stem_rp(IW) + 3) & ~3) + STEM_CHUNK * 128) * 4 <= 160 * 1024 &&
                                       (int64_t)(2 * 1 + 5) * 3 * (IW / 4) <= 16 * 512;
                    if (!oks || !fits1) continue;
                    int rows_max = 1;
                    while (rows_max < o.H && (int64_t)((((2 * (rows_max + 1) + 6) * stem_rp(IW) + 3) & ~3) + STEM_CHUNK * 128) * 4 <= 160 * 1024 &&
                           (int64_t)(2 * (rows_max + 1) + 5) * 3 * (IW / 4) <= 16 * 512) ++rows_max;
                    int wpi = std::max(1, std::min(o.H, 256 / std::max(1, o.N)));        // 256 CUs
                    int rows = (o.H + wpi - 1) / wpi;
                    if (rows > rows_max) rows = rows_max;
                    wpi = (o.H + rows - 1) / rows;
                    n.wg3_S = n.wg3_S16 = o.N * wpi; n.wg3_rows = n.wg3_rows16 = rows; n.wg3_wpi = wpi; n.wg3_entry = n_wgred++;
                    wgred_count[b]++;
                    wgred_maxnumel[b] = std::max(wgred_maxnumel[b], c.Cout * c.Cin);
                    continue;
                }
                if (n.type != N_CONV || n.bucket != b || !enable) continue;
                const ConvInfo& c = convs[n.conv];
                const TensorInfo& o = tensors[n.out];
                if (n.taps == 9) {
                    // 3x3: LDS ring of image rows (wgrad3_3x3_kernel); splits are ranges of image rows
                    const TensorInfo& xi = tensors[n.segs[0].tensor];
                    const bool ok3 = enable3 && c.Cout == 32 && o.ld == 32 && n.Ccat == 128 && n.segs.size() == 1 && !n.segs[0].ups &&
                                     xi.C == 128 && xi.ld % 4 == 0 && o.W >= 2 && o.W >= min_w3 && o.W <= 64 && o.W % 2 == 0 && o.rows() >= min_m;
                    if (!ok3) continue;
                    const int64_t NH = (int64_t)o.N * o.H;
                    int64_t S = (NH + min_rows3 - 1) / min_rows3;
                    if (S > smax) S = smax;
                    if (S < 1) S = 1;
                    const int64_t rows = (NH + S - 1) / S;
                    S = (NH + rows - 1) / rows;
                    n.wg3_S = (int)S; n.wg3_rows = (int)rows; n.wg3_entry = n_wgred++;
                    {
                        int64_t S16 = std::min<int64_t>(S, smax16);
                        int64_t r16 = (NH + S16 - 1) / S16;
                        r16 += r16 & 1;                          // the bf16-MFMA ring kernel walks rows in pairs
                        n.wg3_S16 = (int)((NH + r16 - 1) / r16); n.wg3_rows16 = (int)r16;
                    }
                    wgred_count[b]++;
                    wgred_maxnumel[b] = std::max(wgred_maxnumel[b], c.Cout * n.Ccat * 9);
                    continue;
                }
                // (never a heat-map head: with heads_on_side its weight gradient is enqueued at the START of backward, ahead of
                // the earlier-processed buckets' kernels that reuse the same partial region -- a class_num == 128 head would
                // have its partial tiles overwritten before its bucket's reduce)
                bool ok = n.head < 0 && c.Cout == 128 && o.ld == 128 && n.Ccat % 32 == 0 && n.Ccat >= 128 && o.rows() >= min_m;
                for (auto& sr : n.segs) ok = ok && tensors[sr.tensor].C % 4 == 0 && tensors[sr.tensor].ld % 4 == 0;
                if (!ok) continue;
                const int64_t M = o.rows();
                int64_t S = (M + (int64_t)P * min_chunks - 1) / ((int64_t)P * min_chunks);
                if (S > smax) S = smax;
                if (S < 1) S = 1;
                int64_t rows = (M + S - 1) / S;
                rows = (rows + 63) / 64 * 64;                // whole chunks of both kernels (32 fp32 / 64 bf16 pixels)
                S = (M + rows - 1) / rows;
                n.wg3_S = (int)S; n.wg3_rows = (int)rows; n.wg3_entry = n_wgred++;
                // the fused data + weight gradient (fp32, conv_body's XBG = 4 loop) writes one partial tile per row block of its launch: at most
                // two 4-wave blocks per CU over >= 4 column slices, never more than there are 32-row tiles
                n.fuse_ok = po.fuse_wgrad ? 1 : 0;
                n.wg3_cap = n.fuse_ok ? (int)std::max<int64_t>(S, std::min<int64_t>((M + 31) / 32, 256)) : (int)S;
                {
                    int64_t S16 = (M + (int64_t)P * min_chunks16 - 1) / ((int64_t)P * min_chunks16);
                    S16 = std::max<int64_t>(1, std::min<int64_t>(std::min<int64_t>(S16, smax16), S));
                    int64_t r16 = (M + S16 - 1) / S16;
                    r16 = (r16 + 63) / 64 * 64;
                    n.wg3_S16 = (int)((M + r16 - 1) / r16); n.wg3_rows16 = (int)r16;
                }
                wgred_count[b]++;
                wgred_maxnumel[b] = std::max(wgred_maxnumel[b], c.Cout * n.Ccat);
            }
        }
    }
    off_wgred_tab = off;
    off += round_up64((int64_t)3 * (n_wgred > 0 ? n_wgred : 1) * (int64_t)sizeof(WgReduceEntry), 256);      // one copy per storage mode (runtime.hip, cunet_bind)
    // QuanInput sites (3x3 convs and heads): bit-mask storage for the AND-popcount forward of the quantised-input mode
    {
        int64_t words = 0;
        n_tern_sites = 0;
        for (auto& n : nodes) {
            if (n.type != N_CONV || !(n.taps == 9 || n.head >= 0)) continue;
            ConvInfo& c = convs[n.conv];
            const int G = (c.Cin + 63) / 64;
            if (G > (n.taps == 9 ? 2 : 6)) continue;           // the kernel keeps the masks in registers
            c.tern = words;
            words += 2 * (int64_t)c.taps * G * round_up(c.Cout, 64);
            ++n_tern_sites;
        }
        off_ternpack_tab = off;
        off += round_up64((int64_t)(n_tern_sites > 0 ? n_tern_sites : 1) * (int64_t)sizeof(TernPackEntry), 256);
        off_tern = off;
        off += round_up64(words * 8, 256);
        int64_t max_rows = 0;
        for (auto& n : nodes)
            if (n.type == N_CONV && convs[n.conv].tern >= 0) max_rows = std::max(max_rows, tensors[n.segs[0].tensor].rows());
        off_planes = off;
        off += round_up64((max_rows + 1) * TERN_REC_WORDS * 8, 256);
    }
    loss_acc = n_zero_doubles;
    n_zero_doubles += 2;
    off_zero = off;
    zero_bytes = round_up64(n_zero_doubles * 8, 256);
    off += zero_bytes;
    off_floats = off;

    int64_t f = 0;
    auto take = [&](int64_t n) { int64_t o = f; f += round_up64(n, 64); return o; };
    for (auto& c : convs) {
        c.wF = take((int64_t)c.taps * c.KpadF * c.NpadF);
        if (c.wB >= 0) c.wB = take((int64_t)c.taps * c.KpadB * c.NpadB);
    }
    for (auto& t : tensors) t.act = take(t.rows() * t.ld);
    n_floats_infer = f;
    int64_t dzmax = 0;
    for (auto& n : nodes)
        if (n.type == N_CONV) {
            const TensorInfo& o = tensors[n.out];
            dzmax = std::max(dzmax, o.rows() * (int64_t)n.Ccat);
        }
    for (auto& t : tensors) t.grad = take(t.rows() * t.ld);
    dz_off = f;
    for (auto& n : nodes)
        if (n.type == N_CONV) n.dz = take(tensors[n.out].rows() * (int64_t)n.Ccat);
    (void)dzmax;
    const TensorInfo& h0 = tensors[head_tensors[0]];
    target_off = take(h0.rows() * h0.ld);
    {   // partial tiles of the wgrad3 nodes: TWO regions, used alternately by the buckets in the order backward visits them (round 4).
        // The side stream that runs a bucket's weight gradients also runs its reduce, so side-stream partials never overtake a reduce;
        // but the fused data + weight gradient of the fp32 1x1 nodes writes its partials from the CALLER's stream: with one region the
        // next bucket's data gradients would overwrite tiles the previous bucket's reduce is still reading.  With two, the caller's
        // stream only has to wait for the reduce of the bucket before the previous one (runtime.hip, red_ev).
        int64_t region = 0;
        for (int b = 0; b <= cfg.layer_num; ++b) {
            int64_t sum = 0;
            for (auto& n : nodes)
                if (n.wg3_S > 0 && n.bucket == b) {
                    if (n.wg3_cap < n.wg3_S) n.wg3_cap = n.wg3_S;
                    n.wg3_part = sum;
                    sum += round_up64((int64_t)n.wg3_cap * wg3_numel(n), 64);
                }
            region = std::max(region, sum);
        }
        wg3_region = region;
        const int64_t base = take(2 * region);
        for (auto& n : nodes)
            if (n.wg3_S > 0) n.wg3_part += base + (bucket_position(n.bucket) & 1) * region;
    }
    n_floats_train = f;
    ws_bytes_infer = off_floats + n_floats_infer * 4;
    ws_bytes_train = off_floats + n_floats_train * 4;
    off_bf16 = round_up64(ws_bytes_infer, 256);
    ws_bytes_bf16 = off_bf16 + n_floats_infer * 2;
    off_bf16_train = round_up64(ws_bytes_train, 256);
    ws_bytes_bf16_train = off_bf16_train + n_floats_infer * 2;
}

void Plan::describe() {
    std::ostringstream o;
    o << "{\"cfg\":{\"neck_size\":" << cfg.neck_size << ",\"growth_rate\":" << cfg.growth_rate
      << ",\"init_chan_num\":" << cfg.init_chan_num << ",\"class_num\":" << cfg.class_num
      << ",\"layer_num\":" << cfg.layer_num << ",\"order\":" << cfg.order << ",\"loss_num\":" << cfg.loss_num
      << ",\"batch\":" << cfg.batch << ",\"height\":" << cfg.height << ",\"width\":" << cfg.width << "},";
    o << "\"anchors\":[";
    for (size_t i = 0; i < anchors.size(); ++i) o << (i ? "," : "") << anchors[i];
    o << "],\"off_floats\":" << off_floats << ",\"dz\":" << dz_off << ",\"tensors\":[";
    for (size_t i = 0; i < tensors.size(); ++i) {
        const TensorInfo& t = tensors[i];
        o << (i ? "," : "") << "{\"id\":" << i << ",\"name\":\"" << t.name << "\",\"N\":" << t.N << ",\"H\":" << t.H
          << ",\"W\":" << t.W << ",\"C\":" << t.C << ",\"ld\":" << t.ld << ",\"gld16\":" << t.gld16 << ",\"act\":" << t.act
          << ",\"grad\":" << t.grad << ",\"stats\":" << t.stats << "}";
    }
    o << "],\"nodes\":[";
    for (size_t i = 0; i < nodes.size(); ++i) {
        const Node& n = nodes[i];
        static const char* tn[] = {"stem_conv", "stem_bnpool", "conv", "pool"};
        o << (i ? "," : "") << "{\"op\":\"" << tn[n.type] << "\",\"name\":\"" << n.name << "\",\"out\":" << n.out;
        if (n.bn >= 0) o << ",\"bn\":\"" << bns[n.bn].name << "\",\"ckpt\":" << (bns[n.bn].ckpt ? 1 : 0);
        if (n.conv >= 0) o << ",\"conv\":\"" << convs[n.conv].name << "\",\"taps\":" << n.taps;
        o << ",\"head\":" << n.head << ",\"wg3\":" << n.wg3_S << ",\"wg3_rows\":" << n.wg3_rows << ",\"wg3_bf16\":" << n.wg3_S16
          << ",\"wg3_rows_bf16\":" << n.wg3_rows16 << ",\"wg3_wpi\":" << n.wg3_wpi << ",\"wg3_part\":" << n.wg3_part
          << ",\"wg3_numel\":" << (n.wg3_S > 0 ? wg3_numel(n) : 0) << ",\"fuse_wgrad\":" << n.fuse_ok << ",\"wg3_cap\":" << n.wg3_cap << ",\"bucket\":" << n.bucket << ",\"pair\":" << n.pair << ",\"segs\":[";
        for (size_t s = 0; s < n.segs.size(); ++s)
            o << (s ? "," : "") << "{\"t\":" << n.segs[s].tensor << ",\"ups\":" << n.segs[s].ups
              << "}";
        o << "]}";
    }
    o << "]}";
    json = o.str();
}

}  // namespace cunet
