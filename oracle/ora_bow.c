/*
 * oracle/ora_bow.c -- CPU restatement of the bag-of-words vocabulary walk and of match::bow_tree.  TEST INFRASTRUCTURE ONLY (ora.h).
 *
 * [UPSTREAM] DBoW2 TemplatedVocabulary<TDescriptor,F>::transform(feature, word_id, weight, nid, levelsup) and
 * transform(features, BowVector, FeatureVector, levelsup) (shinsumicco/DBoW2 @ e8cc74d,
 * /root/reference/conan-packages/dbow2-conan/conanfile.py:30-31); [UPSTREAM] openvslam match::bow_tree::match_frame_and_keyframe.
 * The reference needs the vocabulary to start (/root/reference/src/Trackers/OpenVSLAMTrackerBase.cpp:224-227).
 * Written as the serial loops of those functions -- vectors of child ids, one query after the other -- not as the device kernels.
 */
#include "ora.h"
#include <stdlib.h>
#include <string.h>

static int hamming32(const uint8_t* a, const uint8_t* b)
{
    int d = 0;
    for (int i = 0; i < 32; ++i) d += __builtin_popcount((unsigned)(a[i] ^ b[i]));
    return d;
}

/* nodes 1 .. n_nodes in file order (parent[i - 1] = parent of node i, 0 = root); desc / weight / is_leaf likewise.
 * For every descriptor: the word (leaf counted in node order), its weight and the node at level L - levels_up (0 = root). */
int ora_bow_transform(const int32_t* parent, const uint8_t* node_desc, const float* weight, const uint8_t* is_leaf, int n_nodes, int L,
                      const uint8_t* desc, int n, int levels_up, int32_t* word_id, float* word_weight, int32_t* node_id)
{
    /* children of every node, in file order (m_nodes[parent].children.push_back(nid)) */
    int* n_child = (int*)calloc((size_t)n_nodes + 1, sizeof(int));
    int* start = (int*)calloc((size_t)n_nodes + 2, sizeof(int));
    int* list = (int*)calloc((size_t)n_nodes + 1, sizeof(int));
    int* fill = (int*)calloc((size_t)n_nodes + 1, sizeof(int));
    int* word = (int*)calloc((size_t)n_nodes + 1, sizeof(int));
    if (!n_child || !start || !list || !fill || !word) return -1;
    for (int i = 1; i <= n_nodes; ++i) n_child[parent[i - 1]]++;
    for (int i = 0; i <= n_nodes; ++i) start[i + 1] = start[i] + n_child[i];
    for (int i = 1; i <= n_nodes; ++i) { const int p = parent[i - 1]; list[start[p] + fill[p]++] = i; }
    int n_words = 0;
    for (int i = 1; i <= n_nodes; ++i) word[i] = is_leaf[i - 1] ? n_words++ : -1;
    const int nid_level = L - levels_up;
    for (int f = 0; f < n; ++f) {
        const uint8_t* q = desc + 32 * (size_t)f;
        int nid = 0, final_id = 0, current_level = 0;
        do {
            ++current_level;
            const int c0 = start[final_id], c1 = start[final_id + 1];
            final_id = list[c0];
            int best_d = hamming32(q, node_desc + 32 * (size_t)(final_id - 1));
            for (int c = c0 + 1; c < c1; ++c) {
                const int id = list[c];
                const int d = hamming32(q, node_desc + 32 * (size_t)(id - 1));
                if (d < best_d) { best_d = d; final_id = id; }
            }
            if (current_level == nid_level) nid = final_id;
        } while (!is_leaf[final_id - 1]);
        word_id[f] = word[final_id];
        word_weight[f] = weight[final_id - 1];
        node_id[f] = nid;
    }
    free(n_child); free(start); free(list); free(fill); free(word);
    return n_words;
}

typedef struct { int32_t node, idx; } NodeEntry;
static int cmp_entry(const void* a, const void* b)
{
    const NodeEntry* x = (const NodeEntry*)a; const NodeEntry* y = (const NodeEntry*)b;
    if (x->node != y->node) return x->node < y->node ? -1 : 1;
    return x->idx < y->idx ? -1 : (x->idx > y->idx ? 1 : 0);
}

/* match::bow_tree: the two feature vectors (node -> keypoint indices, both sorted) are walked together; under a common node every
 * query (q_node >= 0) looks for its nearest and second nearest free target.  Returns the number of matches. */
int ora_bow_tree_match(const uint8_t* q_desc, const int32_t* q_node, int nq, const uint8_t* t_desc, const int32_t* t_node, int nt,
                       const uint8_t* t_taken_in, int hamming_thr, float lowe_ratio, int32_t* match_idx, int32_t* match_dist)
{
    NodeEntry* qs = (NodeEntry*)malloc(sizeof(NodeEntry) * (size_t)(nq > 0 ? nq : 1));
    NodeEntry* ts = (NodeEntry*)malloc(sizeof(NodeEntry) * (size_t)(nt > 0 ? nt : 1));
    uint8_t* taken = (uint8_t*)calloc((size_t)(nt > 0 ? nt : 1), 1);
    int nqs = 0, nts = 0, found = 0;
    for (int k = 0; k < nq; ++k) { match_idx[k] = -1; if (match_dist) match_dist[k] = 256; if (q_node[k] >= 0) { qs[nqs].node = q_node[k]; qs[nqs++].idx = k; } }
    for (int t = 0; t < nt; ++t) { if (t_taken_in && t_taken_in[t]) taken[t] = 1; if (t_node[t] >= 0) { ts[nts].node = t_node[t]; ts[nts++].idx = t; } }
    qsort(qs, (size_t)nqs, sizeof(NodeEntry), cmp_entry);
    qsort(ts, (size_t)nts, sizeof(NodeEntry), cmp_entry);
    int qi = 0, ti = 0;
    while (qi < nqs && ti < nts) {
        if (qs[qi].node < ts[ti].node) { ++qi; continue; }
        if (ts[ti].node < qs[qi].node) { ++ti; continue; }
        const int node = qs[qi].node;
        int q_end = qi, t_end = ti;
        while (q_end < nqs && qs[q_end].node == node) ++q_end;
        while (t_end < nts && ts[t_end].node == node) ++t_end;
        for (int a = qi; a < q_end; ++a) {
            const int k = qs[a].idx;
            int best = 256, second = 256, best_t = -1;
            for (int b = ti; b < t_end; ++b) {
                const int t = ts[b].idx;
                if (taken[t]) continue;
                const int d = hamming32(q_desc + 32 * (size_t)k, t_desc + 32 * (size_t)t);
                if (d < best) { second = best; best = d; best_t = t; }
                else if (d < second) second = d;
            }
            if (best_t < 0) continue;
            if (hamming_thr < best) continue;
            if (lowe_ratio * (float)second < (float)best) continue;
            taken[best_t] = 1;
            match_idx[k] = best_t;
            if (match_dist) match_dist[k] = best;
            ++found;
        }
        qi = q_end; ti = t_end;
    }
    free(qs); free(ts); free(taken);
    return found;
}
