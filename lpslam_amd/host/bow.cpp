#include "bow.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <sstream>

namespace LpSlam {

bool Vocabulary::load(const std::string& path, std::string& error)
{
    parent.clear(); desc.clear(); weight.clear(); is_leaf.clear();
    std::ifstream f(path, std::ios::binary);
    if (!f) { error = "cannot open " + path; return false; }
    std::vector<char> raw((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    if (raw.size() < 24) { error = "vocabulary file too short"; return false; }
    // binary: the second header word is the record size (4 + 32 + 4 + 1); a text file starts with digits and blanks
    uint32_t hdr[6];
    memcpy(hdr, raw.data(), sizeof(hdr));
    // The node count of the header: the DBoW2 lineage writes m_nodes.size(), which counts the ROOT, and then one record for each of
    // the nodes 1 .. size-1 (the pinned fork's source is absent); files that count only the records (this repo's
    // round-3 trainer) are read as well.  What decides is the file itself: records = (size - 24) / 41.
    const size_t records = (raw.size() - 24) / 41;
    if (hdr[1] == 41 && records > 0 && ((size_t)hdr[0] == records + 1 || (size_t)hdr[0] == records)) {
        k = (int)hdr[2]; L = (int)hdr[3]; scoring = (int)hdr[4]; weighting = (int)hdr[5];
        if (k < 2 || k > 64 || L < 1 || L > 10) { error = "vocabulary header: branching factor / depth out of range"; return false; }
        const size_t n = records;
        parent.resize(n); desc.resize(n * 32); weight.resize(n); is_leaf.resize(n);
        const char* p = raw.data() + 24;
        for (size_t i = 0; i < n; ++i, p += 41) {
            uint32_t par; float w;
            memcpy(&par, p, 4); memcpy(&desc[i * 32], p + 4, 32); memcpy(&w, p + 36, 4);
            if (par > i) { error = "node " + std::to_string(i + 1) + " names a parent that does not precede it"; return false; }
            parent[i] = (int32_t)par; weight[i] = w; is_leaf[i] = p[40] ? 1 : 0;
        }
        return n > 0;
    }
    std::istringstream ss(std::string(raw.begin(), raw.end()));
    if (!(ss >> k >> L >> scoring >> weighting) || k < 2 || k > 64 || L < 1 || L > 10) { error = "not a DBoW2 vocabulary (neither the binary nor the text layout)"; return false; }
    for (;;) {
        long par; int leaf;
        if (!(ss >> par >> leaf)) break;
        uint8_t d[32];
        for (int b = 0; b < 32; ++b) { int x; if (!(ss >> x)) { error = "truncated node record"; return false; } d[b] = (uint8_t)x; }
        double w;
        if (!(ss >> w)) { error = "truncated node record"; return false; }
        if (par < 0 || par > (long)parent.size()) { error = "node names a parent that does not precede it"; return false; }
        parent.push_back((int32_t)par); desc.insert(desc.end(), d, d + 32); weight.push_back((float)w); is_leaf.push_back(leaf ? 1 : 0);
    }
    if (parent.empty()) { error = "vocabulary without nodes"; return false; }
    return true;
}

bool Vocabulary::save_binary(const std::string& path) const
{
    std::ofstream f(path, std::ios::binary);
    if (!f) return false;
    const uint32_t hdr[6] = {(uint32_t)parent.size() + 1u /* nodes incl. the root, as DBoW2 counts them */, 41u, (uint32_t)k, (uint32_t)L, (uint32_t)scoring, (uint32_t)weighting};
    f.write((const char*)hdr, sizeof(hdr));
    for (size_t i = 0; i < parent.size(); ++i) {
        const uint32_t par = (uint32_t)parent[i];
        f.write((const char*)&par, 4); f.write((const char*)&desc[i * 32], 32); f.write((const char*)&weight[i], 4);
        const char leaf = (char)is_leaf[i];
        f.write(&leaf, 1);
    }
    return (bool)f;
}

BowVector make_bow_vector(const int32_t* word_id, const float* word_weight, int n)
{
    // by word, descriptors of a word in their own order (the order of the additions): keys word << 32 | index
    // (a stable sort of pairs through a comparator was 80 us of a keyframe's insertion)
    std::vector<uint64_t> keys, tmp;
    keys.reserve((size_t)n);
    uint32_t top = 0;
    for (int i = 0; i < n; ++i) if (word_weight[i] > 0) { keys.push_back(((uint64_t)(uint32_t)word_id[i] << 32) | (uint32_t)i); top |= (uint32_t)word_id[i]; }
    // LSD radix sort on the word, 11 bits a pass (stable: the index order inside a word survives); std::sort of 2000 keys was 50 us
    tmp.resize(keys.size());
    for (int shift = 32; shift < 64 && (top >> (shift - 32)) != 0; shift += 11) {
        uint32_t hist[2049] = {0};
        for (uint64_t k : keys) hist[((k >> shift) & 2047u) + 1]++;
        for (int b = 0; b < 2048; ++b) hist[b + 1] += hist[b];
        for (uint64_t k : keys) tmp[hist[(k >> shift) & 2047u]++] = k;
        keys.swap(tmp);
    }
    BowVector out;
    out.reserve(keys.size());
    for (uint64_t k : keys) {
        const int32_t w = (int32_t)(uint32_t)(k >> 32);
        const double x = (double)word_weight[(size_t)(k & 0xffffffffu)];
        if (!out.empty() && out.back().first == w) out.back().second += x; else out.emplace_back(w, x);
    }
    double s = 0;
    for (auto& e : out) s += std::fabs(e.second);
    if (s > 0) for (auto& e : out) e.second /= s;
    return out;
}

double bow_score_l1(const BowVector& a, const BowVector& b)
{
    double s = 0;
    size_t i = 0, j = 0;
    while (i < a.size() && j < b.size()) {
        if (a[i].first == b[j].first) { s += std::fabs(a[i].second - b[j].second) - std::fabs(a[i].second) - std::fabs(b[j].second); ++i; ++j; }
        else if (a[i].first < b[j].first) ++i; else ++j;
    }
    return -s / 2.0;
}

void BowDatabase::add(int kf, const BowVector& v)
{
    m_vec[kf] = v;
    for (auto& e : v) m_inv[e.first].push_back(kf);
}

void BowDatabase::remove(int kf)
{
    auto it = m_vec.find(kf);
    if (it == m_vec.end()) return;
    for (auto& e : it->second) {
        auto pl = m_inv.find(e.first);
        if (pl == m_inv.end()) continue;
        auto& v = pl->second;
        v.erase(std::remove(v.begin(), v.end(), kf), v.end());
    }
    m_vec.erase(it);
}

std::vector<std::pair<double, int>> BowDatabase::query(const BowVector& q, const std::unordered_map<int, char>& exclude, double min_score, int max_kf_id) const
{
    std::unordered_map<int, int> common;
    for (auto& e : q) {
        auto it = m_inv.find(e.first);
        if (it == m_inv.end()) continue;
        for (int kf : it->second) if (kf <= max_kf_id && !exclude.count(kf)) common[kf]++;
    }
    int max_common = 0;
    for (auto& kv : common) max_common = std::max(max_common, kv.second);
    std::vector<std::pair<double, int>> out;
    if (max_common == 0) return out;
    const double min_common = 0.8 * max_common;
    for (auto& kv : common) {
        if (kv.second < min_common) continue;
        const double sc = bow_score_l1(q, m_vec.at(kv.first));
        if (min_score >= 0 && sc < min_score) continue;
        out.emplace_back(sc, kv.first);
    }
    std::sort(out.begin(), out.end(), [](const std::pair<double, int>& a, const std::pair<double, int>& b) { return a.first != b.first ? a.first > b.first : a.second > b.second; });
    return out;
}

}  // namespace LpSlam

// C shim for the tests (ctypes): file round trip and the vector arithmetic
extern "C" {
__attribute__((visibility("default"))) int lpslam_bow_vocab_load(const char* path, int32_t* k, int32_t* L, int32_t* n_nodes, int32_t* parent, uint8_t* desc, float* weight, uint8_t* is_leaf, int32_t capacity)
{
    LpSlam::Vocabulary v; std::string err;
    if (!v.load(path, err)) return -1;
    if (k) *k = v.k;
    if (L) *L = v.L;
    if (n_nodes) *n_nodes = v.nodes();
    if (capacity < v.nodes()) return v.nodes();
    if (parent) memcpy(parent, v.parent.data(), v.parent.size() * 4);
    if (desc) memcpy(desc, v.desc.data(), v.desc.size());
    if (weight) memcpy(weight, v.weight.data(), v.weight.size() * 4);
    if (is_leaf) memcpy(is_leaf, v.is_leaf.data(), v.is_leaf.size());
    return v.nodes();
}
__attribute__((visibility("default"))) double lpslam_bow_score(const int32_t* wa, const float* xa, int na, const int32_t* wb, const float* xb, int nb)
{
    return LpSlam::bow_score_l1(LpSlam::make_bow_vector(wa, xa, na), LpSlam::make_bow_vector(wb, xb, nb));
}
}
