// bow.h -- host side of the bag-of-words place recognition: vocabulary files, BoW vectors, the keyframe database.
//
// [UPSTREAM] DBoW2 (shinsumicco/DBoW2 @ e8cc74d, /root/reference/conan-packages/dbow2-conan/conanfile.py:30-31) TemplatedVocabulary
// file formats, BowVector (TF-IDF weights, L1 norm), L1Scoring; [UPSTREAM] openvslam data::bow_database
// (acquire_loop_candidates / acquire_relocalization_candidates).  The reference needs the vocabulary file to start
// (/root/reference/src/Trackers/OpenVSLAMTrackerBase.cpp:224-227).  The tree walk itself runs on the device (csrc/bow.hip).
#pragma once
#include <cstdint>
#include <string>
#include <unordered_map>
#include <utility>
#include <vector>

namespace LpSlam {

struct Vocabulary {
    int k = 0, L = 0, scoring = 0, weighting = 0;
    std::vector<int32_t> parent;       // node i (1-based, file order) -> parent (0 = root)
    std::vector<uint8_t> desc;         // 32 bytes per node
    std::vector<float> weight;
    std::vector<uint8_t> is_leaf;
    int nodes() const { return (int)parent.size(); }
    // Reads a vocabulary: the binary layout of TemplatedVocabulary::loadFromBinaryFile (header nb_nodes, size_node, k, L, scoring,
    // weighting as uint32; per node: parent uint32, 32 descriptor bytes, weight float32, is_leaf uint8) or the text layout of
    // loadFromTextFile ("k L scoring weighting", then per node "parent is_leaf d0 .. d31 weight").
    bool load(const std::string& path, std::string& error);
    bool save_binary(const std::string& path) const;
};

// sparse BoW vector, word ids ascending
typedef std::vector<std::pair<int32_t, double>> BowVector;
// TF-IDF weights of the keypoints' words added up per word, then L1-normalised (TemplatedVocabulary::transform + BowVector::normalize)
BowVector make_bow_vector(const int32_t* word_id, const float* word_weight, int n);
// L1Scoring::score: -sum over common words (|a - b| - |a| - |b|) / 2, in [0, 1]
double bow_score_l1(const BowVector& a, const BowVector& b);

// inverted index word -> keyframes that contain it
class BowDatabase {
public:
    void add(int kf, const BowVector& v);
    void remove(int kf);
    void clear() { m_inv.clear(); m_vec.clear(); }
    const BowVector* vector_of(int kf) const { auto it = m_vec.find(kf); return it == m_vec.end() ? nullptr : &it->second; }
    // keyframes that share words with `q`, without those in `exclude`: the ones with at least 0.8 x the largest number of common
    // words are scored (L1) and returned best first; min_score < 0: no threshold ([UPSTREAM] bow_database::
    // acquire_relocalization_candidates / acquire_loop_candidates, without the covisibility-group accumulation)
    std::vector<std::pair<double, int>> query(const BowVector& q, const std::unordered_map<int, char>& exclude, double min_score, int max_kf_id) const;
private:
    std::unordered_map<int32_t, std::vector<int>> m_inv;
    std::unordered_map<int, BowVector> m_vec;
};

}  // namespace LpSlam
