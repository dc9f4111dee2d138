// bow.hip -- bag-of-words vocabulary on the device: descriptor -> word (tree descent) and match::bow_tree.
//
// [UPSTREAM] DBoW2 TemplatedVocabulary<ORB> (shinsumicco/DBoW2 @ e8cc74d, /root/reference/conan-packages/dbow2-conan/conanfile.py:30-31),
// which openvslam::system loads at start-up -- the reference refuses to start without the file
// (/root/reference/src/Trackers/OpenVSLAMTrackerBase.cpp:224-227, handed to openvslam::system at :238) -- and [UPSTREAM]
// openvslam match::bow_tree (match_frame_and_keyframe / match_keyframes), used by the relocaliser and the loop detector that the
// reference toggles at :250-255.  Neither source is in /root/reference; what is restated here is the published algorithm:
//   transform   from the root, at every level the child with the smallest Hamming distance (the first one on ties); the leaf is the
//               word, its weight the word's (tf-)idf weight, and the node `levels_up` levels above the leaves groups keypoints for
//               matching (FeatureVector);
//   bow_tree    keypoints of the two sides that fall under the same node are compared: per query (in node order, then keypoint
//               order) the nearest free target, accepted at <= hamming_thr and best < ratio * second; a matched target is
//               invisible to later queries.
// Kernels: k_bow_transform (one lane = one descriptor: the walk is a dependent chain of K 32-byte gathers per level, 2000
// descriptors = 32 wavefronts; latency bound, ~L round trips), k_bow_topk (one wavefront per query over the targets of its node, 4
// best per query); the order-dependent part (a target taken by an earlier query) is replayed on the host over those short lists,
// as the window matchers do (match.hip).
#include <chrono>
#include "internal.h"
#include <algorithm>
#include <cstring>
#include <vector>

using namespace lpslam;

struct lpslam_hip_vocab {
    lpslam_hip_ctx* ctx = nullptr;
    int k = 0, L = 0, n_nodes = 0, n_words = 0;      // n_nodes without the root (node ids 1 .. n_nodes)
    // device arrays indexed by node id (0 = root)
    int32_t* d_child_start = nullptr;                // [n_nodes + 2]: children of node i = child_list[child_start[i] .. child_start[i + 1])
    int32_t* d_child_list = nullptr;                 // [n_nodes]
    uint8_t* d_desc = nullptr;                       // [n_nodes + 1][32]
    float* d_weight = nullptr;                       // [n_nodes + 1]
    int32_t* d_word = nullptr;                       // [n_nodes + 1]: word id of a leaf, -1 otherwise
    void* block = nullptr;
};

namespace {

// [UPSTREAM] DBoW2 TemplatedVocabulary::transform: every descriptor walks down the tree to its word, at each node to the child with the
// smallest Hamming distance (the FIRST child on a tie); node_id = the ancestor `levels_up` levels above the leaves (FeatureVector).
// SIXTEEN lanes per descriptor: a lane takes every sixteenth child of the node, the nearest child is the minimum of (distance, child
// order) over the sixteen lanes, so a level costs one round of loads instead of k in a row (k = 10, L = 3 .. 6: 30 .. 60 dependent loads
// with a thread per descriptor -- the kernel a keyframe's insertion waited for, 180 us with its four copies back).
// Results go straight into page-locked host memory [count | word ids | weights | node ids], the last store releases `seq` into *flag.
__global__ __launch_bounds__(256) void k_bow_transform16(const uint8_t* __restrict__ desc, const int32_t* __restrict__ count, int n_fixed,
                                                         const int32_t* __restrict__ child_start, const int32_t* __restrict__ child_list,
                                                         const uint8_t* __restrict__ node_desc, const float* __restrict__ node_weight,
                                                         const int32_t* __restrict__ node_word, int L, int levels_up, int cap,
                                                         int32_t* __restrict__ out_count, int32_t* __restrict__ word_id, float* __restrict__ word_weight, int32_t* __restrict__ node_id,
                                                         unsigned* counter, int* flag, int seq)
{
    const int n = min(count ? *count : n_fixed, cap);
    const int i = (blockIdx.x * 256 + threadIdx.x) >> 4, sub = threadIdx.x & 15;
    if (i < n) {
        const uint4* q4 = reinterpret_cast<const uint4*>(desc + 32 * (size_t)i);
        const uint4 qa = q4[0], qb = q4[1];
        const int nid_level = L - levels_up;
        int nid = 0, cur = 0, level = 0;
        for (;;) {
            const int c0 = child_start[cur], c1 = child_start[cur + 1];
            if (c0 >= c1) break;                           // a leaf (the same for the sixteen lanes of a descriptor)
            ++level;
            unsigned best = 0xffffffffu;                   // distance << 16 | child order
            for (int c = c0 + sub; c < c1; c += 16) {
                const int id = child_list[c];
                const uint4* d4 = reinterpret_cast<const uint4*>(node_desc + 32 * (size_t)id);
                const uint4 da = d4[0], db = d4[1];
                const int d = __popc(qa.x ^ da.x) + __popc(qa.y ^ da.y) + __popc(qa.z ^ da.z) + __popc(qa.w ^ da.w) +
                              __popc(qb.x ^ db.x) + __popc(qb.y ^ db.y) + __popc(qb.z ^ db.z) + __popc(qb.w ^ db.w);
                best = min(best, ((unsigned)d << 16) | (unsigned)(c - c0));
            }
#pragma unroll
            for (int o = 8; o > 0; o >>= 1) best = min(best, (unsigned)__shfl_xor((int)best, o, 16));
            cur = child_list[c0 + (int)(best & 0xffffu)];
            if (level == nid_level) nid = cur;
        }
        if (sub == 0) { word_id[i] = node_word[cur]; word_weight[i] = node_weight[cur]; node_id[i] = nid; }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) *out_count = n;
    lp_signal_done(counter, flag, seq);
}

__device__ __forceinline__ unsigned long long wave_min_u64b(unsigned long long v)
{
    for (int o = 32; o > 0; o >>= 1) { const unsigned long long t = __shfl_xor(v, o); v = t < v ? t : v; }
    return v;
}

// one wavefront per query: the 4 nearest free targets among positions [seg_lo, seg_hi) of the node-sorted target order;
// key = distance << 32 | position (position order = target keypoint order inside a node: the first minimum wins)
__device__ __forceinline__ void bow_topk_body(int bx, const uint8_t* __restrict__ q_desc, const int2* __restrict__ q_seg, const int* __restrict__ q_ids, int nq,
                                                  const uint8_t* __restrict__ t_desc, const int32_t* __restrict__ t_order, const uint8_t* __restrict__ t_taken,
                                                  unsigned long long* __restrict__ out_keys, int* __restrict__ out_count)
{
    const int lane = threadIdx.x & 63, qslot = bx * 4 + (threadIdx.x >> 6);
    if (qslot >= nq) return;
    const int qi = q_ids ? q_ids[qslot] : qslot;
    const int2 seg = q_seg[qi];
    const uint32_t* qd = reinterpret_cast<const uint32_t*>(q_desc + 32 * (size_t)qi);
    uint32_t a[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) a[k] = qd[k];
    const unsigned long long NONE = ~0ull;
    unsigned long long top[4] = {NONE, NONE, NONE, NONE};
    int cnt = 0;
    for (int s = seg.x + lane; s < seg.y; s += 64) {
        const int t = t_order ? t_order[s] : s;          // (no order table: the targets were sent down in node order)
        if (t_taken[t]) continue;
        const uint32_t* d = reinterpret_cast<const uint32_t*>(t_desc + 32 * (size_t)t);
        int dist = 0;
#pragma unroll
        for (int w = 0; w < 8; ++w) dist += __popc(a[w] ^ d[w]);
        unsigned long long key = ((unsigned long long)dist << 32) | (unsigned)s;
        ++cnt;
#pragma unroll
        for (int r = 0; r < 4; ++r) if (key < top[r]) { const unsigned long long tmp = top[r]; top[r] = key; key = tmp; }
    }
    for (int o2 = 32; o2 > 0; o2 >>= 1) cnt += __shfl_xor(cnt, o2);
    unsigned long long res[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const unsigned long long m = wave_min_u64b(top[0]);
        res[r] = m;
        if (top[0] == m && m != NONE) { top[0] = top[1]; top[1] = top[2]; top[2] = top[3]; top[3] = NONE; }
    }
    if (lane == 0) {
#pragma unroll
        for (int r = 0; r < 4; ++r) out_keys[4 * (size_t)qslot + r] = res[r];
        out_count[qslot] = cnt;
    }
}

__global__ __launch_bounds__(256) void k_bow_topk(const uint8_t* __restrict__ q_desc, const int2* __restrict__ q_seg, const int* __restrict__ q_ids, int nq,
                                                  const uint8_t* __restrict__ t_desc, const int32_t* __restrict__ t_order, const uint8_t* __restrict__ t_taken,
                                                  unsigned long long* __restrict__ out_keys, int* __restrict__ out_count)
{
    bow_topk_body(blockIdx.x, q_desc, q_seg, q_ids, nq, t_desc, t_order, t_taken, out_keys, out_count);
}
// several target sets in one launch: blockIdx.y = set, its arrays at byte offsets of one block (targets already in node order)
struct BowSetOff { unsigned seg, td, taken, keys, cnt, pad[3]; };
__global__ __launch_bounds__(256) void k_bow_topk_sets(const uint8_t* __restrict__ base, const BowSetOff* __restrict__ sets, int nq)
{
    const BowSetOff o = sets[blockIdx.y];
    bow_topk_body(blockIdx.x, base, (const int2*)(base + o.seg), (const int*)nullptr, nq, base + o.td, (const int32_t*)nullptr, base + o.taken,
                  (unsigned long long*)(const_cast<uint8_t*>(base) + o.keys), (int*)(const_cast<uint8_t*>(base) + o.cnt));
}

}  // namespace

extern "C" {

int lpslam_hip_vocab_create(lpslam_hip_ctx* c, int32_t k, int32_t L, int32_t n_nodes, const int32_t* parent, const uint8_t* desc32, const float* weight,
                            const uint8_t* is_leaf, lpslam_hip_vocab** out)
{
    if (!c || !parent || !desc32 || !weight || !is_leaf || !out || n_nodes < 1 || k < 2 || L < 1) { set_error("invalid vocabulary arguments"); return LPSLAM_HIP_ERR_INVALID; }
    *out = nullptr;
    // nodes arrive as DBoW2 stores them: node i (1-based) names its parent, which precedes it; children keep that order
    std::vector<int32_t> n_child((size_t)n_nodes + 1, 0);
    for (int i = 1; i <= n_nodes; ++i) {
        const int p = parent[i - 1];
        if (p < 0 || p >= i) { set_error("vocabulary node %d names parent %d (must precede it)", i, p); return LPSLAM_HIP_ERR_INVALID; }
        n_child[(size_t)p]++;
    }
    std::vector<int32_t> start((size_t)n_nodes + 2, 0), list((size_t)n_nodes, 0), fill((size_t)n_nodes + 1, 0), word((size_t)n_nodes + 1, -1);
    for (int i = 0; i <= n_nodes; ++i) start[(size_t)i + 1] = start[(size_t)i] + n_child[(size_t)i];
    for (int i = 1; i <= n_nodes; ++i) { const int p = parent[i - 1]; list[(size_t)(start[(size_t)p] + fill[(size_t)p]++)] = i; }
    int n_words = 0;
    for (int i = 1; i <= n_nodes; ++i) {
        const bool leaf = n_child[(size_t)i] == 0;
        if (leaf != (is_leaf[i - 1] != 0)) { set_error("vocabulary node %d: leaf flag does not match its children", i); return LPSLAM_HIP_ERR_INVALID; }
        if (leaf) word[(size_t)i] = n_words++;          // word ids in node order (DBoW2: m_words in file order)
    }
    if (n_child[0] == 0) { set_error("the vocabulary's root has no children"); return LPSLAM_HIP_ERR_INVALID; }
    LP_HIP(hipSetDevice(c->cfg.device));
    lpslam_hip_vocab* v = new lpslam_hip_vocab();
    v->ctx = c; v->k = k; v->L = L; v->n_nodes = n_nodes; v->n_words = n_words;
    const size_t nn = (size_t)n_nodes + 1;
    auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
    const size_t o_start = 0, o_list = o_start + al((nn + 1) * 4), o_desc = o_list + al(nn * 4), o_w = o_desc + al(nn * 32), o_word = o_w + al(nn * 4), total = o_word + al(nn * 4);
    if (hipMalloc(&v->block, total) != hipSuccess) { delete v; set_error("hipMalloc of %zu bytes for the vocabulary failed", total); return LPSLAM_HIP_ERR_DEVICE; }
    std::vector<uint8_t> h(total, 0);
    memcpy(h.data() + o_start, start.data(), (nn + 1) * 4);
    memcpy(h.data() + o_list, list.data(), (size_t)n_nodes * 4);
    memcpy(h.data() + o_desc + 32, desc32, (size_t)n_nodes * 32);          // node 0 (the root) has no descriptor
    memcpy(h.data() + o_w + 4, weight, (size_t)n_nodes * 4);
    memcpy(h.data() + o_word, word.data(), nn * 4);
    if (hipMemcpy(v->block, h.data(), total, hipMemcpyHostToDevice) != hipSuccess) { (void)hipFree(v->block); delete v; set_error("hipMemcpy failed"); return LPSLAM_HIP_ERR_DEVICE; }
    uint8_t* b = (uint8_t*)v->block;
    v->d_child_start = (int32_t*)(b + o_start); v->d_child_list = (int32_t*)(b + o_list); v->d_desc = b + o_desc; v->d_weight = (float*)(b + o_w); v->d_word = (int32_t*)(b + o_word);
    *out = v;
    return LPSLAM_HIP_OK;
}

void lpslam_hip_vocab_destroy(lpslam_hip_vocab* v)
{
    if (!v) return;
    if (v->ctx) { (void)hipSetDevice(v->ctx->cfg.device); (void)hipStreamSynchronize(v->ctx->stream); }
    if (v->block) (void)hipFree(v->block);
    delete v;
}

int lpslam_hip_vocab_info(lpslam_hip_vocab* v, int32_t* k, int32_t* L, int32_t* n_nodes, int32_t* n_words)
{
    if (!v) { set_error("null vocabulary"); return LPSLAM_HIP_ERR_INVALID; }
    if (k) *k = v->k;
    if (L) *L = v->L;
    if (n_nodes) *n_nodes = v->n_nodes;
    if (n_words) *n_words = v->n_words;
    return LPSLAM_HIP_OK;
}

static int bow_transform_device(lpslam_hip_ctx* c, lpslam_hip_vocab* v, const uint8_t* d_desc, const int32_t* d_count, int n_max, int levels_up,
                                int32_t* word_id, float* word_weight, int32_t* node_id, int32_t* count_out)
{
    hipStream_t s = c->stream;
    const size_t nm = (size_t)std::max(n_max, 1);
    if (v->k > 65535) { set_error("vocabulary: more than 65535 children per node"); return LPSLAM_HIP_ERR_CAPACITY; }
    // page-locked block: done flag | count | word ids | weights | node ids -- written by the kernel itself, released by its last store
    const size_t o_word = 64, o_w = o_word + nm * 4, o_node = o_w + nm * 4, total = o_node + nm * 4;
    if (c->h_match_bytes < total) {
        if (c->h_match) { LP_HIP(hipStreamSynchronize(s)); (void)hipHostFree(c->h_match); }
        c->h_match = nullptr; c->h_match_bytes = 0;
        LP_HIP(hipHostMalloc((void**)&c->h_match, total + total / 2, hipHostMallocDefault));
        c->h_match_bytes = total + total / 2;
    }
    uint8_t* hb = c->h_match;
    unsigned* counter = lp_done_counter(c, 4);
    if (!counter) { set_error("device memory for the completion counters"); return LPSLAM_HIP_ERR_DEVICE; }
    int* flag = (int*)hb;
    const int seq = lp_next_seq(c->done_seq);
    __atomic_store_n(flag, 0, __ATOMIC_RELAXED);
    hipLaunchKernelGGL(k_bow_transform16, dim3((unsigned)((nm * 16 + 255) / 256)), dim3(256), 0, s, d_desc, d_count, n_max, v->d_child_start, v->d_child_list, v->d_desc, v->d_weight,
                       v->d_word, v->L, levels_up, n_max, (int32_t*)(hb + 32), (int32_t*)(hb + o_word), (float*)(hb + o_w), (int32_t*)(hb + o_node), counter, flag, seq);
    LP_HIP(hipGetLastError());
    if (!lp_wait_done(flag, seq, s)) { (void)lp_wait_recover(c, 4, s); set_error("bow_transform: the kernel did not complete"); return LPSLAM_HIP_ERR_DEVICE; }
    const int n = std::min(std::max(*(const int32_t*)(hb + 32), 0), n_max);
    if (n > 0) {
        if (word_id) memcpy(word_id, hb + o_word, (size_t)n * 4);
        if (word_weight) memcpy(word_weight, hb + o_w, (size_t)n * 4);
        if (node_id) memcpy(node_id, hb + o_node, (size_t)n * 4);
    }
    if (count_out) *count_out = n;
    return LPSLAM_HIP_OK;
}

int lpslam_hip_bow_transform(lpslam_hip_ctx* c, lpslam_hip_vocab* v, int image, int32_t levels_up, int32_t* word_id, float* word_weight, int32_t* node_id,
                             int32_t capacity, int32_t* count)
{
    if (!c || !v || v->ctx != c) { set_error("null context / vocabulary of another context"); return LPSLAM_HIP_ERR_INVALID; }
    if (image < 0 || image >= c->cfg.max_images) { set_error("image slot %d out of range [0,%d)", image, c->cfg.max_images); return LPSLAM_HIP_ERR_CAPACITY; }
    if (capacity < c->slots_per_image) { set_error("result buffers too small (%d < %d)", capacity, c->slots_per_image); return LPSLAM_HIP_ERR_CAPACITY; }
    LP_HIP(hipSetDevice(c->cfg.device));
    const size_t o = (size_t)image * c->slots_per_image;
    return bow_transform_device(c, v, c->d_desc + o * 32, c->d_kp_count + image, c->slots_per_image, levels_up, word_id, word_weight, node_id, count);
}

int lpslam_hip_bow_transform_host(lpslam_hip_ctx* c, lpslam_hip_vocab* v, const uint8_t* desc32, int32_t n, int32_t levels_up, int32_t* word_id, float* word_weight,
                                  int32_t* node_id)
{
    if (!c || !v || v->ctx != c || n < 0 || (n > 0 && !desc32)) { set_error("invalid bow_transform arguments"); return LPSLAM_HIP_ERR_INVALID; }
    if (n == 0) return LPSLAM_HIP_OK;
    LP_HIP(hipSetDevice(c->cfg.device));
    void* d = nullptr; size_t cap = 0;
    { const int rc = lp_pool_alloc(c, (size_t)n * 32, &d, &cap); if (rc) return rc; }
    if (hipMemcpyAsync(d, desc32, (size_t)n * 32, hipMemcpyHostToDevice, c->stream) != hipSuccess) { lp_pool_free(c, d, cap); set_error("hipMemcpy failed"); return LPSLAM_HIP_ERR_DEVICE; }
    const int rc = bow_transform_device(c, v, (const uint8_t*)d, nullptr, n, levels_up, word_id, word_weight, node_id, nullptr);
    lp_pool_free(c, d, cap);
    return rc;
}

// One query set against n_sets target sets in ONE round trip: inputs staged contiguously in page-locked memory and sent down with one
// copy, a k_bow_topk launch per set, the candidate lists back with one copy and one wait; the order-dependent part (a target matched by
// an earlier query is invisible to later ones) is replayed on the host per set, with the single-query re-scan where a short list was
// eaten.  (A loop-candidate search matches the new keyframe against up to eight keyframes: eight calls were eight uploads and eight waits,
// 47 us each.)
int lpslam_hip_match_bow_tree_multi(lpslam_hip_ctx* c, const uint8_t* q_desc32, const int32_t* q_node, int32_t nq, int32_t n_sets, const uint8_t* const* t_desc32,
                                    const int32_t* const* t_node, const int32_t* nt_of, const uint8_t* const* t_taken_in, int32_t hamming_thr, float lowe_ratio,
                                    int32_t* const* match_idx, int32_t* const* match_dist, int32_t* n_matches)
{
    if (!c || nq < 0 || n_sets < 0 || (n_sets > 0 && (!t_desc32 || !t_node || !nt_of || !match_idx)) || (nq > 0 && n_sets > 0 && (!q_desc32 || !q_node))) { set_error("invalid bow_tree arguments"); return LPSLAM_HIP_ERR_INVALID; }
    for (int i = 0; i < n_sets; ++i) {
        if (nt_of[i] < 0 || (nq > 0 && !match_idx[i]) || (nt_of[i] > 0 && (!t_desc32[i] || !t_node[i]))) { set_error("invalid bow_tree arguments (set %d)", i); return LPSLAM_HIP_ERR_INVALID; }
        if (n_matches) n_matches[i] = 0;
        for (int k = 0; k < nq; ++k) { match_idx[i][k] = -1; if (match_dist && match_dist[i]) match_dist[i][k] = 256; }
    }
    if (nq == 0 || n_sets == 0) return LPSLAM_HIP_OK;
    LP_HIP(hipSetDevice(c->cfg.device));
    hipStream_t s = c->stream;
    static const bool trace = getenv("LPSLAM_HIP_MATCH_TRACE") != nullptr;
    const auto tr0 = std::chrono::steady_clock::now();
    auto tr_us = [&tr0]() { return 1e-3 * (double)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - tr0).count(); };
    double tr_staged = 0, tr_back = 0;
    // the order upstream visits the queries in: node by node (ascending id), keypoint order inside a node
    std::vector<int32_t> q_order;
    for (int k = 0; k < nq; ++k) if (q_node[k] >= 0) q_order.push_back(k);
    std::stable_sort(q_order.begin(), q_order.end(), [&](int32_t a, int32_t b) { return q_node[a] < q_node[b]; });
    if (q_order.empty()) return LPSLAM_HIP_OK;
    // Only what takes part travels: the queries that have a node, in the order they are served (query a = q_order[a]), and per set the
    // targets that have a node, in node order (target position j = t_order[j]; a loop search's keyframes carry landmarks on a quarter of
    // their keypoints: 2 x 16 KB per set instead of 2 x 64 KB down, 16 KB of lists instead of 64 KB back).
    const int nqa = (int)q_order.size();
    struct Set { std::vector<int32_t> t_order; int nto = 0; size_t o_seg = 0, o_td = 0, o_taken = 0, o_keys = 0, o_cnt = 0; bool live = false; };
    std::vector<Set> sets((size_t)n_sets);
    auto al = [](size_t x) { return (x + 63) & ~(size_t)63; };
    // block: [query descriptors | per set: segments, target descriptors, taken]  (inputs, one copy down)  [per set: keys, counts]  (one copy back)  [ids]
    size_t at = al((size_t)nqa * 32);
    for (int i = 0; i < n_sets; ++i) {
        Set& st = sets[(size_t)i];
        const int nt = nt_of[i];
        st.t_order.reserve((size_t)nt);
        for (int t = 0; t < nt; ++t) if (t_node[i][t] >= 0) st.t_order.push_back(t);
        std::stable_sort(st.t_order.begin(), st.t_order.end(), [&](int32_t a, int32_t b) { return t_node[i][a] < t_node[i][b]; });
        st.nto = (int)st.t_order.size();
        st.live = st.nto > 0;
        if (!st.live) continue;
        st.o_seg = at; at += al((size_t)nqa * 8);
        st.o_td = at; at += al((size_t)st.nto * 32);
        st.o_taken = at; at += al((size_t)st.nto);
    }
    const size_t o_tab = at; at += al((size_t)n_sets * sizeof(BowSetOff));      // offsets of the live sets (k_bow_topk_sets)
    const size_t in_end = at;
    for (Set& st : sets) if (st.live) { st.o_keys = at; at += al((size_t)nqa * 4 * 8); st.o_cnt = at; at += al((size_t)nqa * 4); }
    const size_t out_end = at, o_ids = at, total = o_ids + 64;
    if (o_tab == al((size_t)nqa * 32)) return LPSLAM_HIP_OK;                 // no set has a target under any node
    if (at + 64 >= (size_t)0xffffffffu) { set_error("bow_tree: the sets do not fit one block"); return LPSLAM_HIP_ERR_CAPACITY; }
    void* blk = nullptr; size_t cap = 0;
    { const int rc = lp_pool_alloc(c, total, &blk, &cap); if (rc) return rc; }
    auto release = [&]() { lp_pool_free(c, blk, cap); };
#define B_HIP(x) do { if ((x) != hipSuccess) { release(); set_error("HIP call failed: %s", #x); return LPSLAM_HIP_ERR_DEVICE; } } while (0)
    if (c->h_match_bytes < total) {
        if (c->h_match) { B_HIP(hipStreamSynchronize(s)); (void)hipHostFree(c->h_match); }
        c->h_match = nullptr; c->h_match_bytes = 0;
        B_HIP(hipHostMalloc((void**)&c->h_match, total + total / 2, hipHostMallocDefault));
        c->h_match_bytes = total + total / 2;
    }
    uint8_t* base = (uint8_t*)blk; uint8_t* hb = c->h_match;
    for (int a = 0; a < nqa; ++a) memcpy(hb + 32 * (size_t)a, q_desc32 + 32 * (size_t)q_order[(size_t)a], 32);
    std::vector<std::vector<uint8_t>> taken_of((size_t)n_sets);              // by target index, as the replay keeps it
    for (int i = 0; i < n_sets; ++i) {
        Set& st = sets[(size_t)i];
        if (!st.live) continue;
        const int nt = nt_of[i];
        // segment of every served query in the node-sorted target order: both sides ascend by node, one merge
        int2* seg = (int2*)(hb + st.o_seg);
        int lo = 0;
        for (int a = 0; a < nqa; ) {
            const int node = q_node[q_order[(size_t)a]];
            while (lo < st.nto && t_node[i][st.t_order[(size_t)lo]] < node) ++lo;
            int hi = lo;
            while (hi < st.nto && t_node[i][st.t_order[(size_t)hi]] == node) ++hi;
            for (; a < nqa && q_node[q_order[(size_t)a]] == node; ++a) seg[a] = make_int2(lo, hi);
            lo = hi;
        }
        std::vector<uint8_t>& taken = taken_of[(size_t)i];
        taken.assign((size_t)nt, 0);
        const uint8_t* tin = t_taken_in ? t_taken_in[i] : nullptr;
        if (tin) for (int t = 0; t < nt; ++t) taken[(size_t)t] = tin[t] ? 1 : 0;
        uint8_t* td = hb + st.o_td; uint8_t* tk = hb + st.o_taken;
        for (int j = 0; j < st.nto; ++j) { const int t = st.t_order[(size_t)j]; memcpy(td + 32 * (size_t)j, t_desc32[i] + 32 * (size_t)t, 32); tk[j] = taken[(size_t)t]; }
    }
    int n_live = 0;
    {
        BowSetOff* tab = (BowSetOff*)(hb + o_tab);
        for (Set& st : sets) if (st.live) tab[n_live++] = BowSetOff{(unsigned)st.o_seg, (unsigned)st.o_td, (unsigned)st.o_taken, (unsigned)st.o_keys, (unsigned)st.o_cnt, {0, 0, 0}};
    }
    tr_staged = tr_us();
    B_HIP(hipMemcpyAsync(base, hb, in_end, hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(k_bow_topk_sets, dim3((unsigned)((nqa + 3) / 4), (unsigned)n_live), dim3(256), 0, s, (const uint8_t*)base, (const BowSetOff*)(base + o_tab), nqa);
    B_HIP(hipGetLastError());
    B_HIP(hipMemcpyAsync(hb + in_end, base + in_end, out_end - in_end, hipMemcpyDeviceToHost, s));
    B_HIP(hipStreamSynchronize(s));
    tr_back = tr_us();
    int* d_ids = (int*)(base + o_ids);
    int found_all = 0;
    for (int i = 0; i < n_sets; ++i) {
        Set& st = sets[(size_t)i];
        if (!st.live) continue;
        const unsigned long long* keys = (const unsigned long long*)(hb + st.o_keys);
        const int* cnt = (const int*)(hb + st.o_cnt);
        std::vector<uint8_t>& taken = taken_of[(size_t)i];
        const std::vector<int32_t>& t_order = st.t_order;
        int found = 0;
        for (int a = 0; a < nqa; ++a) {
            const int32_t k = q_order[(size_t)a];
            unsigned long long cand[4] = {keys[4 * (size_t)a], keys[4 * (size_t)a + 1], keys[4 * (size_t)a + 2], keys[4 * (size_t)a + 3]};
            auto free_ones = [&](unsigned long long* out) {
                int m = 0;
                for (int r = 0; r < 4; ++r) if (cand[r] != ~0ull && !taken[(size_t)t_order[(size_t)(cand[r] & 0xffffffffu)]]) out[m++] = cand[r];
                return m; };
            unsigned long long fr[4];
            int m = free_ones(fr);
            if (m < 2 && cnt[a] > 4) {
                // the short list was eaten by earlier queries: scan again for this query with the current assignment
                uint8_t* tk = hb + st.o_taken;
                for (int j = 0; j < st.nto; ++j) tk[j] = taken[(size_t)t_order[(size_t)j]];
                B_HIP(hipMemcpyAsync(base + st.o_taken, tk, (size_t)st.nto, hipMemcpyHostToDevice, s));
                B_HIP(hipMemcpyAsync(d_ids, &a, sizeof(int), hipMemcpyHostToDevice, s));
                hipLaunchKernelGGL(k_bow_topk, dim3(1), dim3(256), 0, s, base, (const int2*)(base + st.o_seg), (const int*)d_ids, 1, base + st.o_td,
                                   (const int32_t*)nullptr, base + st.o_taken, (unsigned long long*)(base + st.o_keys), (int*)(base + st.o_cnt));
                B_HIP(hipGetLastError());
                B_HIP(hipMemcpyAsync(cand, base + st.o_keys, 4 * sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
                B_HIP(hipStreamSynchronize(s));
                m = free_ones(fr);
            }
            if (m == 0) continue;
            const int best = (int)(fr[0] >> 32), best_t = t_order[(size_t)(fr[0] & 0xffffffffu)];
            const int second = m > 1 ? (int)(fr[1] >> 32) : 256;
            if (hamming_thr < best) continue;
            if (lowe_ratio * (float)second < (float)best) continue;
            taken[(size_t)best_t] = 1;
            match_idx[i][k] = best_t;
            if (match_dist && match_dist[i]) match_dist[i][k] = best;
            ++found;
        }
        if (n_matches) n_matches[i] = found;
        found_all += found;
    }
#undef B_HIP
    release();
    if (trace) fprintf(stderr, "bow_tree_match: %d queries, %d targets, %d matches; us: sorted %.1f, staged %.1f, lists back %.1f, replayed %.1f\n", nq, n_sets, found_all, 0.0, tr_staged, tr_back, tr_us());
    return LPSLAM_HIP_OK;
}

int lpslam_hip_match_bow_tree(lpslam_hip_ctx* c, const uint8_t* q_desc32, const int32_t* q_node, int32_t nq, const uint8_t* t_desc32, const int32_t* t_node, int32_t nt,
                              const uint8_t* t_taken_in, int32_t hamming_thr, float lowe_ratio, int32_t* match_idx, int32_t* match_dist, int32_t* n_matches)
{
    if (!c || nq < 0 || nt < 0 || (nq > 0 && (!q_desc32 || !q_node || !match_idx)) || (nt > 0 && (!t_desc32 || !t_node))) { set_error("invalid bow_tree arguments"); return LPSLAM_HIP_ERR_INVALID; }
    return lpslam_hip_match_bow_tree_multi(c, q_desc32, q_node, nq, 1, &t_desc32, &t_node, &nt, t_taken_in ? &t_taken_in : nullptr, hamming_thr, lowe_ratio, &match_idx, &match_dist, n_matches);
}

}  // extern "C"
