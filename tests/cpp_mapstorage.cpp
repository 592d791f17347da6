// Driver of include/se2lam_amd/MapStorage.h for tests/test_mapstorage.py (host only, no device):
//   cpp_mapstorage gen <dir/> <seed> <nkf> <nmp>     build a deterministic map (the generator is mirrored in the test),
//                                                    MapStorage::saveMap into dir + the key-frame trajectory
//   cpp_mapstorage copy <dir_in/> <dir_out/>         MapStorage::loadMap from dir_in, saveMap into dir_out (round trip)
//   cpp_mapstorage text <dir_in/> <file>             loadMap, then the YAML text only (no images) into <file>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "se2lam_amd/MapStorage.h"

using namespace se2lam_amd;

// xorshift64*: trivially reproducible in Python
struct Rng {
    uint64_t s;
    explicit Rng(uint64_t seed) : s(seed * 2654435761ull + 88172645463325252ull) {}
    uint64_t next() { s ^= s >> 12; s ^= s << 25; s ^= s >> 27; return s * 2685821657736338717ull; }
    int below(int n) { return (int)((next() >> 33) % (uint64_t)n); }
    float unit() { return (float)((next() >> 40) * (1.0 / 16777216.0)); }   // 24 bits: exact in float
};

static StoredMap generate(uint64_t seed, int nkf, int nmp) {
    Rng r(seed);
    StoredMap m;
    for (int j = 0; j < nmp; ++j) {
        StoredMapPoint mp;
        mp.mId = 1000 + j;
        mp.null = r.below(10) == 0;
        mp.goodPrl = r.below(8) != 0;
        mp.pos = Point3f{r.unit() * 8000.f - 4000.f, r.unit() * 3000.f, (float)r.below(5000)};   // integers among them: "123."
        m.mps.push_back(mp);
    }
    for (int i = 0; i < nkf; ++i) {
        StoredKeyFrame kf;
        kf.id = 7 * i + 3;
        kf.mIdKF = 50 + i;
        kf.null = (i % 5) == 3;
        const int n = i == 1 ? 0 : 3 + r.below(6);            // one key frame without key points: "[]" of an empty structure
        kf.descriptors = StoredMat::zeros(n, 32, 'u');
        for (int k = 0; k < n; ++k) {
            KeyPoint kp;
            kp.pt.x = r.unit() * 640.f; kp.pt.y = (float)r.below(480);
            kp.octave = r.below(8);
            kp.angle = r.unit() * 360.f;
            kp.response = (float)r.below(200);
            kf.keyPoints.push_back(kp);
            kp.pt.x += 0.25f;
            kf.keyPointsUn.push_back(kp);
            for (int b = 0; b < 32; ++b) kf.descriptors.u[(size_t)k * 32 + b] = (uint8_t)r.below(256);
            kf.mViewMPs.push_back(Point3f{r.unit() * 100.f, -r.unit(), r.below(3) ? r.unit() * 9000.f : -1.f});
            Matrix3D I;
            for (int q = 0; q < 9; ++q) I.m[q] = (q % 4 == 0) ? 1.0 / (1 + r.below(50)) : (double)r.unit() * 1e-3;
            kf.mViewMPsInfo.push_back(I);
        }
        // a planar body pose (x, y, theta) seen through Config::bTc: Tcw = (Twb Tbc)^-1, as KeyFrame poses are in se2lam
        const float th = r.unit() * 6.28f - 3.14f, bx = r.unit() * 5000.f, by = r.unit() * -5000.f;
        {
            const double Rbc[9] = {0, 0, 1, -1, 0, 0, 0, -1, 0}, tbc[3] = {100, 0, 300};
            const double c = std::cos((double)th), sn = std::sin((double)th);
            const double Rwb[9] = {c, -sn, 0, sn, c, 0, 0, 0, 1}, twb[3] = {bx, by, 0};
            double Rwc[9], twc[3];
            for (int a = 0; a < 3; ++a) {
                for (int b = 0; b < 3; ++b) {
                    Rwc[3 * a + b] = 0;
                    for (int k = 0; k < 3; ++k) Rwc[3 * a + b] += Rwb[3 * a + k] * Rbc[3 * k + b];
                }
                twc[a] = twb[a];
                for (int k = 0; k < 3; ++k) twc[a] += Rwb[3 * a + k] * tbc[k];
            }
            for (int a = 0; a < 3; ++a) {
                double t = 0;
                for (int b = 0; b < 3; ++b) {
                    kf.Tcw.at<float>(a, b) = (float)Rwc[3 * b + a];
                    t -= Rwc[3 * b + a] * twc[b];
                }
                kf.Tcw.at<float>(a, 3) = (float)t;
            }
        }
        kf.odom[0] = r.unit() * 1000.f; kf.odom[1] = r.unit() * 1000.f; kf.odom[2] = th;
        kf.mfScaleFactor = 1.2f;
        const int rows = 5 + r.below(4), cols = 6 + r.below(5);   // widths that need row padding in the bitmap
        kf.img.create(rows, cols);
        for (int y = 0; y < rows; ++y)
            for (int x = 0; x < cols; ++x) kf.img.ptr(y)[x] = (uint8_t)r.below(256);
        for (int k = 0; k < n; ++k)
            if (nmp && r.below(2)) kf.observations.emplace_back(r.below(nmp), k);
        // (covisibility is mutual in a real map and loadMap adds it both ways: made symmetric below; an odometry edge points
        // at a key frame that survives saveMap - entries with NextId < 0 are skipped by loadOdoGraph, MapStorage.cpp:547-548)
        for (int c = 0; c < i; ++c)
            if (r.below(3) == 0) kf.covisible.push_back(c);
        const bool next_alive = i + 1 < nkf && ((i + 1) % 5) != 3;
        if (next_alive && r.below(4)) {
            kf.odoNext = i + 1;
            kf.odoMeasure = StoredMat::zeros(6, 1, 'f');
            for (auto& v : kf.odoMeasure.f) v = r.unit() * 10.f - 5.f;
            kf.odoInfo = StoredMat::zeros(6, 6, 'f');
            for (int q = 0; q < 6; ++q) kf.odoInfo.f[7 * q] = 1.f + r.below(1000);
        }
        const int ne = r.below(3);
        for (int e = 0; e < ne; ++e) {
            StoredFtrEdge fe;
            fe.to = r.below(nkf);
            fe.measure = StoredMat::zeros(6, 1, 'f');
            for (auto& v : fe.measure.f) v = r.unit() - 0.5f;
            fe.info = StoredMat::zeros(6, 6, 'd');
            for (int q = 0; q < 36; ++q) fe.info.d[q] = (q % 7 == 0) ? 1e4 * (1 + r.below(9)) : 1e-7 * r.unit();
            kf.ftrFrom.push_back(fe);
        }
        m.kfs.push_back(std::move(kf));
    }
    for (int i = 0; i < nkf; ++i)
        for (int c : std::vector<int>(m.kfs[i].covisible))
            if (c < i) m.kfs[c].covisible.push_back(i);
    return m;
}

int main(int argc, char** argv) {
    try {
        if (argc >= 6 && !std::strcmp(argv[1], "gen")) {
            StoredMap m = generate((uint64_t)std::atoll(argv[3]), std::atoi(argv[4]), std::atoi(argv[5]));
            MapStorage ms;
            ms.setMap(&m);
            ms.setFilePath(argv[2], "se2lam.map");
            ms.saveMap();
            MatF bTc = MatF::eye(4);
            const float R[9] = {0, 0, 1, -1, 0, 0, 0, -1, 0};
            for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) bTc.at<float>(r, c) = R[3 * r + c];
            bTc.at<float>(0, 3) = 100; bTc.at<float>(2, 3) = 300;
            saveKeyFrameTrajectory(std::string(argv[2]) + "/se2lam_kf_trajectory.txt", m, bTc);
            std::printf("saved %zu key frames, %zu map points\n", m.kfs.size(), m.mps.size());
            return 0;
        }
        if (argc >= 4 && !std::strcmp(argv[1], "copy")) {
            StoredMap m;
            MapStorage ms;
            ms.setMap(&m);
            ms.setFilePath(argv[2], "se2lam.map");
            ms.loadMap();
            std::printf("loaded %zu key frames, %zu map points\n", m.kfs.size(), m.mps.size());
            for (size_t i = 0; i < m.kfs.size(); ++i) {
                const StoredKeyFrame& kf = m.kfs[i];
                std::printf("KF %zu kps %zu obs %zu covis %zu next %d ftr %zu img %dx%d\n", i, kf.keyPoints.size(), kf.observations.size(),
                            kf.covisible.size(), kf.odoNext, kf.ftrFrom.size(), kf.img.rows, kf.img.cols);
            }
            // side channel for the test: what was loaded, in a form that does not pass through the YAML code (hex floats)
            for (size_t i = 0; i < m.kfs.size(); ++i) {
                const StoredKeyFrame& kf = m.kfs[i];
                unsigned long dsum = 0, isum = 0;
                for (uint8_t v : kf.descriptors.u) dsum += v;
                for (int y = 0; y < kf.img.rows; ++y) for (int x = 0; x < kf.img.cols; ++x) isum += kf.img.ptr(y)[x] * (unsigned long)(1 + (x + 3 * y) % 7);
                std::printf("RAW %zu desc %lu img %lu pose3 %a odom %a %a %a scale %a", i, dsum, isum, (double)kf.Tcw.v[3], (double)kf.odom[0],
                            (double)kf.odom[1], (double)kf.odom[2], (double)kf.mfScaleFactor);
                if (!kf.keyPoints.empty()) std::printf(" kp0 %a %a %d %a %a", (double)kf.keyPoints[0].pt.x, (double)kf.keyPoints[0].pt.y,
                                                        kf.keyPoints[0].octave, (double)kf.keyPoints[0].angle, (double)kf.keyPoints[0].response);
                std::printf("\n");
            }
            ms.setFilePath(argv[3], "se2lam.map");
            ms.saveMap();
            for (const TrajectoryEntry& e : loadKeyFrameTrajectory(std::string(argv[2]) + "/se2lam_kf_trajectory.txt"))
                std::printf("TRJ %d %.9g %.9g %.9g %.17g\n", e.id, (double)e.x, (double)e.y, (double)e.z, e.yaw);
            return 0;
        }
    } catch (const std::exception& e) {
        std::fprintf(stderr, "error: %s\n", e.what());
        return 1;
    }
    return 2;
}
