// Drop-in for the one entry point of /root/reference/include/se2lam/sparsifier.h that GlobalMapper calls:
// Sparsifier::DoMarginalizeSE3XYZ (sparsifier.h:54, src/sparsifier.cpp:105-275), same argument meaning, on libse2gpu.
// `DoMarginalizeSE3XYZBatch` is the shape the GPU wants: every key-frame pair of one GlobalMapper::UpdataFeatGraph pass
// (src/GlobalMapper.cpp:117-170 calls CreateFeatEdge once per pair) in ONE launch, one wave per pair.
#pragma once
#include <vector>

#include "../se2gpu.h"
#include "optimizer.h"
#include "types.h"

namespace se2lam_amd {

struct MeasSE3XYZ {            // sparsifier.h:13-19
    Vector3D z;
    Matrix3D info;
    int idMP = -1;
    int idKF = -1;
};

struct FeatPair {              // the arguments of one DoMarginalizeSE3XYZ call
    std::vector<SE3Quat> vKF;  // exactly two key frames, T_w_c (GlobalMapper.cpp:798-800)
    std::vector<Vector3D> vMP;
    std::vector<MeasSE3XYZ> vMeasure;
};

class Sparsifier {
public:
    static void DoMarginalizeSE3XYZBatch(const std::vector<FeatPair>& pairs, std::vector<SE3Quat>& z_out,
                                         std::vector<Matrix6d>& info_out) {
        const int n = (int)pairs.size();
        std::vector<double> kf(24 * (size_t)n), xyz, info, z(12 * (size_t)n), out(36 * (size_t)n);
        std::vector<int32_t> mp_ptr(n + 1, 0), m_ptr(n + 1, 0), m_kf, m_mp;
        for (int p = 0; p < n; ++p) {
            const FeatPair& f = pairs[p];
            if (f.vKF.size() != 2) throw std::runtime_error("DoMarginalizeSE3XYZ: a feature constraint joins exactly two key frames");
            for (int k = 0; k < 2; ++k) pose12Of(f.vKF[k], &kf[24 * (size_t)p + 12 * k]);
            for (const Vector3D& x : f.vMP) xyz.insert(xyz.end(), x.v, x.v + 3);
            for (const MeasSE3XYZ& m : f.vMeasure) {
                m_kf.push_back(m.idKF);
                m_mp.push_back(m.idMP);
                info.insert(info.end(), m.info.m, m.info.m + 9);
            }
            mp_ptr[p + 1] = mp_ptr[p] + (int32_t)f.vMP.size();
            m_ptr[p + 1] = m_ptr[p] + (int32_t)f.vMeasure.size();
        }
        check(se2gpu_sparsify_se3xyz(n, kf.data(), mp_ptr.data(), xyz.data(), m_ptr.data(), m_kf.data(), m_mp.data(),
                                     info.data(), z.data(), out.data()), "DoMarginalizeSE3XYZ");
        z_out.resize(n);
        info_out.resize(n);
        for (int p = 0; p < n; ++p) {
            z_out[p] = se3QuatOf(&z[12 * (size_t)p]);
            for (int i = 0; i < 36; ++i) info_out[p].m[i] = out[36 * (size_t)p + i];
        }
    }
    static void DoMarginalizeSE3XYZ(const std::vector<SE3Quat>& vKF, const std::vector<Vector3D>& vMP,
                                    const std::vector<MeasSE3XYZ>& vMeasure, SE3Quat& z_out, Matrix6d& info_out) {
        std::vector<FeatPair> one(1);
        one[0].vKF = vKF;
        one[0].vMP = vMP;
        one[0].vMeasure = vMeasure;
        std::vector<SE3Quat> z;
        std::vector<Matrix6d> info;
        DoMarginalizeSE3XYZBatch(one, z, info);
        z_out = z[0];
        info_out = info[0];
    }
};

}  // namespace se2lam_amd
