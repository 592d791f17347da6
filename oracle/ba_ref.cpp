// ORACLE - TEST INFRASTRUCTURE ONLY.  Never linked, imported or called by the product path
// (se2lam_amd/, include/).  Only tests/, __graft_entry__.smoke() and bench.py's
// `cpu_baseline` leg may use it, and only as the checker / reported CPU baseline.
//
// PARITY: the two EDGES and the GRAPH CONSTRUCTION - what se2lam itself wrote of this path - are pinned against the
// reference's own code: oracle/_ref compiles /root/reference/src/EdgeSE2XYZ.cpp, src/optimizer.cpp, src/converter.cpp and
// their headers unmodified against a stand-in for the Eigen / g2o headers (oracle/_shim/g2o_shim.hpp), and
// tests/test_ref_compiled.py holds edge_se2xyz / edge_pre_se2 below to their computeError() / linearizeOplus() (1e-11
// relative), the reduced system assembled from the reference's Jacobians to schur() below (1e-10), robust_chi2() to the cost
// of a window built with the reference's add* calls (1e-13), ba_ref_plane_motion_prior to addPlaneMotionSE3Expmap (1e-9),
// and ba_ref_edge_information to the information matrices Map::loadLocalGraph (src/Map.cpp, compiled with KeyFrame.cpp and
// MapPoint.cpp in oracle/_ref/libse2lam_ref_map.so) puts on its edges (1e-10).
// The SOLVER stays UNPINNED: /root/reference holds no golden vectors, known-answer tests or fixtures
// for this path, and g2o / Eigen / CHOLMOD are not available in this image, so the Levenberg-Marquardt histories could
// not be checked against a run of the real reference; they are pinned against scipy instead.
//
// CPU restatement (single thread, FP64, no dependencies) of the SE(2)-XYZ local bundle
// adjustment inner loop of izhengfan/se2lam:
//   * EdgeSE2XYZ::computeError / linearizeOplus   /root/reference/src/EdgeSE2XYZ.cpp:61-106
//   * PreEdgeSE2::computeError / linearizeOplus    /root/reference/include/se2lam/EdgeSE2XYZ.h:62-102
//   * graph semantics of addEdgeSE2XYZ/addVertexSE2/addEdgeSE2/addVertexSBAXYZ/addCamPara
//                                                  /root/reference/src/optimizer.cpp:17-62,207-215,316-325
//   * the solver LocalMapper::localBA builds:  LM( BlockSolverX( LinearSolverCholmod ) )
//                                                  /root/reference/src/LocalMapper.cpp:232-302,
//                                                  /root/reference/include/se2lam/optimizer.h:30-34
// The solver itself lives in the un-vendored dependency g2o (tag 20160424_git, README.MD:29);
// its published algorithm is restated here:
//   - VertexSE2::oplusImpl: (x,y) += d, theta = normalize_theta(theta + dtheta); VertexSBAPointXYZ: additive
//   - RobustKernelHuber::robustify, BaseBinaryEdge::constructQuadraticForm (rho'' term disabled)
//   - BlockSolver::buildSystem / solve with Schur complement on the marginalised landmarks
//   - OptimizationAlgorithmLevenberg::solve: lambda0 = 1e-5*max diag(H), rho test with
//     scale = sum x_i(lambda x_i + b_i) + 1e-3, lambda *= clamp(1-(2rho-1)^3, 1/3, 2/3) on
//     success, lambda *= nu, nu *= 2 on failure, <= 10 trials, Terminate when trials==10 or rho==0
//   - SparseOptimizer::optimize outer loop with the cooperative force-stop flag
//   - CHOLMOD is replaced by a dense LL^T (exact SPD solve; equivalent up to round-off).
//
// Build: g++ -O2 -ffp-contract=off -shared -fPIC (see oracle/Makefile).
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <vector>

extern "C" {

struct ba_ref_problem {
    int32_t P, L, E, O;
    const double* poses;     // P*3  (x,y,theta)
    const uint8_t* fixed;    // P
    const double* lms;       // L*3
    const int32_t* e_kf;     // E
    const int32_t* e_lm;     // E
    const double* e_uv;      // E*2
    const double* e_info;    // E*3  (xx, xy, yy)
    const int32_t* o_i;      // O
    const int32_t* o_j;      // O
    const double* o_meas;    // O*3
    const double* o_info;    // O*9 row-major
    double fx, cx, cy;
    double Rbc[9];           // row-major
    double tbc[3];
    double huber;
};

struct ba_ref_stats {
    int32_t iterations;      // outer iterations executed
    int32_t trials;          // total LM trials (linear solves)
    int32_t terminated;      // 1 if the algorithm returned Terminate
    double chi2_init, chi2_final, lambda_final;
    double chi2_hist[64];    // robust chi2 after each outer iteration
    double lambda_hist[64];
    int32_t trials_hist[64];
    int32_t n_rho;           // gain ratio of every LM trial, in order (first 256): the margin of each accept / reject decision
    double rho_log[256];
};

}  // extern "C"

namespace {

const double kPi = 3.14159265358979323846;

inline double normalize_theta(double theta) {  // g2o/stuff/misc.h
    if (theta >= -kPi && theta < kPi) return theta;
    double multiplier = std::floor(theta / (2 * kPi));
    theta = theta - multiplier * 2 * kPi;
    if (theta >= kPi) theta -= 2 * kPi;
    if (theta < -kPi) theta += 2 * kPi;
    return theta;
}

struct Cam {
    double fx, cx, cy;
    double Rcb[9], tcb[3];
};

Cam make_cam(const ba_ref_problem& p) {
    Cam c;
    c.fx = p.fx; c.cx = p.cx; c.cy = p.cy;
    // Tcb = Tbc^-1 (EdgeSE2XYZ.h:53)
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) c.Rcb[i * 3 + j] = p.Rbc[j * 3 + i];
    for (int i = 0; i < 3; ++i)
        c.tcb[i] = -(c.Rcb[i * 3 + 0] * p.tbc[0] + c.Rcb[i * 3 + 1] * p.tbc[1] + c.Rcb[i * 3 + 2] * p.tbc[2]);
    return c;
}

// EdgeSE2XYZ::computeError + linearizeOplus (src/EdgeSE2XYZ.cpp:61-106), closed form:
// lc = Rcb * Rz(-theta) * (lw - [x,y,0]) + tcb ; e = f*(X/Z, Y/Z) + c - z
inline void edge_se2xyz(const Cam& cam, const double* pose, const double* lw, const double* uv,
                        double e[2], double Jp[6], double Jl[6], bool jac) {
    const double th = pose[2];
    const double c = std::cos(th), s = std::sin(th);
    const double dx = lw[0] - pose[0], dy = lw[1] - pose[1], dz = lw[2];
    // Rcw = Rcb * Rz(-theta),  Rz(-theta) = [[c, s, 0], [-s, c, 0], [0, 0, 1]]
    double Rcw[9];
    for (int i = 0; i < 3; ++i) {
        const double r0 = cam.Rcb[i * 3 + 0], r1 = cam.Rcb[i * 3 + 1], r2 = cam.Rcb[i * 3 + 2];
        Rcw[i * 3 + 0] = r0 * c - r1 * s;
        Rcw[i * 3 + 1] = r0 * s + r1 * c;
        Rcw[i * 3 + 2] = r2;
    }
    double lc[3];
    for (int i = 0; i < 3; ++i)
        lc[i] = Rcw[i * 3 + 0] * dx + Rcw[i * 3 + 1] * dy + Rcw[i * 3 + 2] * dz + cam.tcb[i];
    const double zi = 1.0 / lc[2];
    e[0] = cam.fx * lc[0] * zi + cam.cx - uv[0];
    e[1] = cam.fx * lc[1] * zi + cam.cy - uv[1];
    if (!jac) return;
    const double zi2 = zi * zi;
    const double j00 = cam.fx * zi, j02 = -cam.fx * lc[0] * zi2;
    const double j11 = cam.fx * zi, j12 = -cam.fx * lc[1] * zi2;
    // A = J_pi * Rcw (2x3)
    double A[6];
    for (int k = 0; k < 3; ++k) {
        A[k] = j00 * Rcw[0 * 3 + k] + j02 * Rcw[2 * 3 + k];
        A[3 + k] = j11 * Rcw[1 * 3 + k] + j12 * Rcw[2 * 3 + k];
    }
    // J_pose[:,0:2] = -A[:,0:2];  J_pose[:,2] = (A*skew(lw-pi))[:,2] = A*(dy,-dx,0)
    Jp[0] = -A[0]; Jp[1] = -A[1]; Jp[2] = A[0] * dy - A[1] * dx;
    Jp[3] = -A[3]; Jp[4] = -A[4]; Jp[5] = A[3] * dy - A[4] * dx;
    for (int k = 0; k < 6; ++k) Jl[k] = A[k];
}

// RobustKernelHuber::robustify (g2o/core/robust_kernel_impl.cpp)
inline void huber(double e2, double delta, double& rho0, double& rho1) {
    const double dsqr = delta * delta;
    if (e2 <= dsqr) {
        rho0 = e2; rho1 = 1.0;
    } else {
        const double sqrte = std::sqrt(e2);
        rho0 = 2 * sqrte * delta - dsqr;
        rho1 = delta / sqrte;
    }
}

// PreEdgeSE2 (include/se2lam/EdgeSE2XYZ.h:62-102)
inline void edge_pre_se2(const double* pi, const double* pj, const double* z, double e[3], double Ji[9],
                         double Jj[9], bool jac) {
    const double c = std::cos(pi[2]), s = std::sin(pi[2]);
    const double rx = pj[0] - pi[0], ry = pj[1] - pi[1];
    // Ri^T = [[c, s], [-s, c]]
    e[0] = c * rx + s * ry - z[0];
    e[1] = -s * rx + c * ry - z[1];
    e[2] = pj[2] - pi[2] - z[2];  // no angle wrap
    if (!jac) return;
    const double qx = -ry, qy = rx;  // rij_x
    std::memset(Ji, 0, 9 * sizeof(double));
    std::memset(Jj, 0, 9 * sizeof(double));
    Ji[0] = -c; Ji[1] = -s; Ji[3] = s; Ji[4] = -c;
    Ji[2] = -(c * qx + s * qy);
    Ji[5] = -(-s * qx + c * qy);
    Ji[8] = -1;
    Jj[0] = c; Jj[1] = s; Jj[3] = -s; Jj[4] = c; Jj[8] = 1;
}

struct State {
    std::vector<double> poses, lms;
};

double robust_chi2(const ba_ref_problem& p, const Cam& cam, const State& st) {
    double chi = 0;
    for (int k = 0; k < p.E; ++k) {
        double e[2];
        edge_se2xyz(cam, &st.poses[3 * p.e_kf[k]], &st.lms[3 * p.e_lm[k]], &p.e_uv[2 * k], e, nullptr, nullptr, false);
        const double* w = &p.e_info[3 * k];
        const double e2 = e[0] * (w[0] * e[0] + w[1] * e[1]) + e[1] * (w[1] * e[0] + w[2] * e[1]);
        double r0, r1;
        huber(e2, p.huber, r0, r1);
        chi += r0;
    }
    for (int k = 0; k < p.O; ++k) {
        double e[3];
        edge_pre_se2(&st.poses[3 * p.o_i[k]], &st.poses[3 * p.o_j[k]], &p.o_meas[3 * k], e, nullptr, nullptr, false);
        const double* W = &p.o_info[9 * k];
        double we[3];
        for (int r = 0; r < 3; ++r) we[r] = W[r * 3 + 0] * e[0] + W[r * 3 + 1] * e[1] + W[r * 3 + 2] * e[2];
        chi += e[0] * we[0] + e[1] * we[1] + e[2] * we[2];
    }
    return chi;
}

// Linearised system at the current state (BlockSolver::buildSystem).
struct System {
    int P, L, n;                    // n = 3P
    std::vector<double> Hpp;        // n*n dense (pose-pose incl. odometry blocks), fixed rows/cols zero
    std::vector<double> bp;         // n
    std::vector<double> Hll;        // L*9 (full 3x3 row-major)
    std::vector<double> bl;         // L*3
    std::vector<double> Hpl;        // E*9 row-major 3x3 = Jp^T W Jl (zero if pose fixed)
    std::vector<int> lm_ptr, lm_edges;  // CSR landmark -> edge ids
};

void build_csr(const ba_ref_problem& p, System& sys) {
    sys.lm_ptr.assign(p.L + 1, 0);
    for (int k = 0; k < p.E; ++k) sys.lm_ptr[p.e_lm[k] + 1]++;
    for (int l = 0; l < p.L; ++l) sys.lm_ptr[l + 1] += sys.lm_ptr[l];
    sys.lm_edges.resize(p.E);
    std::vector<int> fill(sys.lm_ptr.begin(), sys.lm_ptr.end() - 1);
    for (int k = 0; k < p.E; ++k) sys.lm_edges[fill[p.e_lm[k]]++] = k;
}

void build_system(const ba_ref_problem& p, const Cam& cam, const State& st, System& sys) {
    const int n = 3 * p.P;
    sys.P = p.P; sys.L = p.L; sys.n = n;
    sys.Hpp.assign((size_t)n * n, 0.0);
    sys.bp.assign(n, 0.0);
    sys.Hll.assign((size_t)p.L * 9, 0.0);
    sys.bl.assign((size_t)p.L * 3, 0.0);
    sys.Hpl.assign((size_t)p.E * 9, 0.0);
    for (int k = 0; k < p.E; ++k) {
        const int kf = p.e_kf[k], lm = p.e_lm[k];
        double e[2], Jp[6], Jl[6];
        edge_se2xyz(cam, &st.poses[3 * kf], &st.lms[3 * lm], &p.e_uv[2 * k], e, Jp, Jl, true);
        const double* w = &p.e_info[3 * k];
        const double we0 = w[0] * e[0] + w[1] * e[1], we1 = w[1] * e[0] + w[2] * e[1];
        const double e2 = e[0] * we0 + e[1] * we1;
        double r0, r1;
        huber(e2, p.huber, r0, r1);
        const double W[3] = {r1 * w[0], r1 * w[1], r1 * w[2]};  // weightedOmega = rho1 * Omega
        const double or0 = -r1 * we0, or1 = -r1 * we1;          // omega_r = -Omega e * rho1
        // W*Jl, W*Jp (2x3)
        double WJl[6], WJp[6];
        for (int c = 0; c < 3; ++c) {
            WJl[c] = W[0] * Jl[c] + W[1] * Jl[3 + c];
            WJl[3 + c] = W[1] * Jl[c] + W[2] * Jl[3 + c];
            WJp[c] = W[0] * Jp[c] + W[1] * Jp[3 + c];
            WJp[3 + c] = W[1] * Jp[c] + W[2] * Jp[3 + c];
        }
        double* Hl = &sys.Hll[(size_t)lm * 9];
        for (int r = 0; r < 3; ++r) {
            for (int c = 0; c < 3; ++c) Hl[r * 3 + c] += Jl[r] * WJl[c] + Jl[3 + r] * WJl[3 + c];
            sys.bl[(size_t)lm * 3 + r] += Jl[r] * or0 + Jl[3 + r] * or1;
        }
        if (!p.fixed[kf]) {
            for (int r = 0; r < 3; ++r) {
                for (int c = 0; c < 3; ++c) {
                    sys.Hpp[(size_t)(3 * kf + r) * n + 3 * kf + c] += Jp[r] * WJp[c] + Jp[3 + r] * WJp[3 + c];
                    sys.Hpl[(size_t)k * 9 + r * 3 + c] = Jp[r] * WJl[c] + Jp[3 + r] * WJl[3 + c];
                }
                sys.bp[3 * kf + r] += Jp[r] * or0 + Jp[3 + r] * or1;
            }
        }
    }
    for (int k = 0; k < p.O; ++k) {
        const int i = p.o_i[k], j = p.o_j[k];
        double e[3], A[9], B[9];
        edge_pre_se2(&st.poses[3 * i], &st.poses[3 * j], &p.o_meas[3 * k], e, A, B, true);
        const double* W = &p.o_info[9 * k];
        double omr[3], WA[9], WB[9];
        for (int r = 0; r < 3; ++r) {
            omr[r] = -(W[r * 3 + 0] * e[0] + W[r * 3 + 1] * e[1] + W[r * 3 + 2] * e[2]);
            for (int c = 0; c < 3; ++c) {
                WA[r * 3 + c] = W[r * 3 + 0] * A[c] + W[r * 3 + 1] * A[3 + c] + W[r * 3 + 2] * A[6 + c];
                WB[r * 3 + c] = W[r * 3 + 0] * B[c] + W[r * 3 + 1] * B[3 + c] + W[r * 3 + 2] * B[6 + c];
            }
        }
        const bool fi = !p.fixed[i], fj = !p.fixed[j];
        for (int r = 0; r < 3; ++r) {
            for (int c = 0; c < 3; ++c) {
                const double AtWA = A[r] * WA[c] + A[3 + r] * WA[3 + c] + A[6 + r] * WA[6 + c];
                const double AtWB = A[r] * WB[c] + A[3 + r] * WB[3 + c] + A[6 + r] * WB[6 + c];
                const double BtWB = B[r] * WB[c] + B[3 + r] * WB[3 + c] + B[6 + r] * WB[6 + c];
                if (fi) sys.Hpp[(size_t)(3 * i + r) * n + 3 * i + c] += AtWA;
                if (fj) sys.Hpp[(size_t)(3 * j + r) * n + 3 * j + c] += BtWB;
                if (fi && fj) {
                    sys.Hpp[(size_t)(3 * i + r) * n + 3 * j + c] += AtWB;
                    sys.Hpp[(size_t)(3 * j + c) * n + 3 * i + r] += AtWB;
                }
            }
            if (fi) sys.bp[3 * i + r] += A[r] * omr[0] + A[3 + r] * omr[1] + A[6 + r] * omr[2];
            if (fj) sys.bp[3 * j + r] += B[r] * omr[0] + B[3 + r] * omr[1] + B[6 + r] * omr[2];
        }
    }
}

inline bool inv3(const double* M, double* Mi) {
    const double a = M[0], b = M[1], c = M[2], d = M[3], e = M[4], f = M[5], g = M[6], h = M[7], i = M[8];
    const double A = e * i - f * h, B = -(d * i - f * g), C = d * h - e * g;
    const double det = a * A + b * B + c * C;
    const double id = 1.0 / det;
    Mi[0] = A * id; Mi[1] = -(b * i - c * h) * id; Mi[2] = (b * f - c * e) * id;
    Mi[3] = B * id; Mi[4] = (a * i - c * g) * id;  Mi[5] = -(a * f - c * d) * id;
    Mi[6] = C * id; Mi[7] = -(a * h - b * g) * id; Mi[8] = (a * e - b * d) * id;
    return std::isfinite(id);
}

// In-place lower Cholesky of the n x n row-major SPD matrix; returns false on a non-positive pivot.
bool cholesky(std::vector<double>& A, int n) {
    for (int j = 0; j < n; ++j) {
        double* Aj = &A[(size_t)j * n];
        double d = Aj[j];
        for (int k = 0; k < j; ++k) d -= Aj[k] * Aj[k];
        if (!(d > 0.0)) return false;
        d = std::sqrt(d);
        Aj[j] = d;
        const double id = 1.0 / d;
        for (int i = j + 1; i < n; ++i) {
            double* Ai = &A[(size_t)i * n];
            double s = Ai[j];
            for (int k = 0; k < j; ++k) s -= Ai[k] * Aj[k];
            Ai[j] = s * id;
        }
    }
    return true;
}

void chol_solve(const std::vector<double>& Lm, int n, std::vector<double>& x) {
    for (int i = 0; i < n; ++i) {
        double s = x[i];
        const double* Li = &Lm[(size_t)i * n];
        for (int k = 0; k < i; ++k) s -= Li[k] * x[k];
        x[i] = s / Li[i];
    }
    for (int i = n - 1; i >= 0; --i) {
        double s = x[i];
        for (int k = i + 1; k < n; ++k) s -= Lm[(size_t)k * n + i] * x[k];
        x[i] = s / Lm[(size_t)i * n + i];
    }
}

// Schur complement for damping lambda (BlockSolver::solve): S, bs (dense, fixed poses -> identity rows).
void schur(const ba_ref_problem& p, const System& sys, double lambda, std::vector<double>& S,
           std::vector<double>& bs, std::vector<double>& Dinv) {
    const int n = sys.n;
    S = sys.Hpp;
    bs = sys.bp;
    for (int i = 0; i < n; ++i) S[(size_t)i * n + i] += lambda;
    Dinv.assign((size_t)p.L * 9, 0.0);
    for (int l = 0; l < p.L; ++l) {
        double D[9];
        std::memcpy(D, &sys.Hll[(size_t)l * 9], sizeof(D));
        D[0] += lambda; D[4] += lambda; D[8] += lambda;
        double* Di = &Dinv[(size_t)l * 9];
        inv3(D, Di);
        const double* b = &sys.bl[(size_t)l * 3];
        const double db[3] = {Di[0] * b[0] + Di[1] * b[1] + Di[2] * b[2], Di[3] * b[0] + Di[4] * b[1] + Di[5] * b[2],
                              Di[6] * b[0] + Di[7] * b[1] + Di[8] * b[2]};
        for (int a = sys.lm_ptr[l]; a < sys.lm_ptr[l + 1]; ++a) {
            const int ea = sys.lm_edges[a];
            const int pa = p.e_kf[ea];
            if (p.fixed[pa]) continue;
            const double* Ba = &sys.Hpl[(size_t)ea * 9];
            double BD[9];
            for (int r = 0; r < 3; ++r)
                for (int c = 0; c < 3; ++c) BD[r * 3 + c] = Ba[r * 3] * Di[c] + Ba[r * 3 + 1] * Di[3 + c] + Ba[r * 3 + 2] * Di[6 + c];
            for (int r = 0; r < 3; ++r) bs[3 * pa + r] -= Ba[r * 3] * db[0] + Ba[r * 3 + 1] * db[1] + Ba[r * 3 + 2] * db[2];
            for (int b2 = sys.lm_ptr[l]; b2 < sys.lm_ptr[l + 1]; ++b2) {
                const int eb = sys.lm_edges[b2];
                const int pb = p.e_kf[eb];
                if (p.fixed[pb]) continue;
                const double* Bb = &sys.Hpl[(size_t)eb * 9];
                for (int r = 0; r < 3; ++r)
                    for (int c = 0; c < 3; ++c)
                        S[(size_t)(3 * pa + r) * n + 3 * pb + c] -=
                            BD[r * 3] * Bb[c * 3] + BD[r * 3 + 1] * Bb[c * 3 + 1] + BD[r * 3 + 2] * Bb[c * 3 + 2];
            }
        }
    }
    for (int k = 0; k < p.P; ++k)
        if (p.fixed[k])
            for (int r = 0; r < 3; ++r) {
                const int i = 3 * k + r;
                for (int c = 0; c < n; ++c) { S[(size_t)i * n + c] = 0; S[(size_t)c * n + i] = 0; }
                S[(size_t)i * n + i] = 1.0;
                bs[i] = 0.0;
            }
}

// x_l = Dinv (b_l - Hpl^T x_p)
void back_substitute(const ba_ref_problem& p, const System& sys, const std::vector<double>& Dinv,
                     const std::vector<double>& xp, std::vector<double>& xl) {
    xl.assign((size_t)p.L * 3, 0.0);
    for (int l = 0; l < p.L; ++l) {
        double c[3] = {sys.bl[(size_t)l * 3], sys.bl[(size_t)l * 3 + 1], sys.bl[(size_t)l * 3 + 2]};
        for (int a = sys.lm_ptr[l]; a < sys.lm_ptr[l + 1]; ++a) {
            const int ea = sys.lm_edges[a];
            const int pa = p.e_kf[ea];
            if (p.fixed[pa]) continue;
            const double* B = &sys.Hpl[(size_t)ea * 9];
            for (int j = 0; j < 3; ++j) c[j] -= B[0 * 3 + j] * xp[3 * pa] + B[1 * 3 + j] * xp[3 * pa + 1] + B[2 * 3 + j] * xp[3 * pa + 2];
        }
        const double* Di = &Dinv[(size_t)l * 9];
        for (int r = 0; r < 3; ++r) xl[(size_t)l * 3 + r] = Di[r * 3] * c[0] + Di[r * 3 + 1] * c[1] + Di[r * 3 + 2] * c[2];
    }
}

void apply_update(const ba_ref_problem& p, const State& in, const std::vector<double>& xp, const std::vector<double>& xl,
                  State& out) {
    out = in;
    for (int k = 0; k < p.P; ++k) {
        if (p.fixed[k]) continue;
        out.poses[3 * k] += xp[3 * k];
        out.poses[3 * k + 1] += xp[3 * k + 1];
        out.poses[3 * k + 2] = normalize_theta(out.poses[3 * k + 2] + xp[3 * k + 2]);
    }
    for (size_t i = 0; i < out.lms.size(); ++i) out.lms[i] += xl[i];
}

}  // namespace

extern "C" {

// error + Jacobians of ONE EdgeSE2XYZ (for analytic-vs-numeric Jacobian tests)
void ba_ref_edge_se2xyz(const ba_ref_problem* p, const double* pose, const double* lw, const double* uv, double* e,
                        double* Jp, double* Jl) {
    Cam cam = make_cam(*p);
    edge_se2xyz(cam, pose, lw, uv, e, Jp, Jl, true);
}

void ba_ref_edge_pre_se2(const double* pi, const double* pj, const double* z, double* e, double* Ji, double* Jj) {
    edge_pre_se2(pi, pj, z, e, Ji, Jj, true);
}

double ba_ref_chi2(const ba_ref_problem* p, const double* poses, const double* lms) {
    Cam cam = make_cam(*p);
    State st;
    st.poses.assign(poses, poses + 3 * p->P);
    st.lms.assign(lms, lms + 3 * (size_t)p->L);
    return robust_chi2(*p, cam, st);
}

// Reduced (Schur) system at the given state and damping: S (3P x 3P row-major), bs (3P),
// plus the un-reduced gradient pieces bp (3P) and bl (3L), Hll (9L).  Any output may be NULL.
void ba_ref_reduced_system(const ba_ref_problem* p, const double* poses, const double* lms, double lambda, double* S,
                           double* bs, double* bp, double* bl, double* Hll) {
    Cam cam = make_cam(*p);
    State st;
    st.poses.assign(poses, poses + 3 * p->P);
    st.lms.assign(lms, lms + 3 * (size_t)p->L);
    System sys;
    build_csr(*p, sys);
    build_system(*p, cam, st, sys);
    std::vector<double> Sv, bsv, Dinv;
    schur(*p, sys, lambda, Sv, bsv, Dinv);
    if (S) std::memcpy(S, Sv.data(), Sv.size() * sizeof(double));
    if (bs) std::memcpy(bs, bsv.data(), bsv.size() * sizeof(double));
    if (bp) std::memcpy(bp, sys.bp.data(), sys.bp.size() * sizeof(double));
    if (bl) std::memcpy(bl, sys.bl.data(), sys.bl.size() * sizeof(double));
    if (Hll) std::memcpy(Hll, sys.Hll.data(), sys.Hll.size() * sizeof(double));
}

// Per-observation information of Map::loadLocalGraph (src/Map.cpp:1024-1049), same argument layout as
// se2gpu_ba_edge_information.  Doubles promoted from the float members exactly where the reference promotes them.
void ba_ref_edge_information(int E, const float* lc_, const float* lw_, const int32_t* e_kf, const float* sigma2, int P,
                             const float* Rcw_, const float* twb_xy, float fx, float xrot_info, float z_info, double* out) {
    (void)P;
    for (int k = 0; k < E; ++k) {
        const double lc[3] = {lc_[3 * k], lc_[3 * k + 1], lc_[3 * k + 2]};
        const double lw[3] = {lw_[3 * k], lw_[3 * k + 1], lw_[3 * k + 2]};
        const int kf = e_kf[k];
        double Rcw[9];
        for (int i = 0; i < 9; ++i) Rcw[i] = Rcw_[9 * kf + i];
        const double pi[3] = {twb_xy[2 * kf], twb_xy[2 * kf + 1], 0.0};
        const double zc = lc[2];
        const double zc_inv = 1. / zc;
        const double zc_inv2 = zc_inv * zc_inv;
        const double Jpi[6] = {fx * zc_inv, 0, -fx * lc[0] * zc_inv2, 0, fx * zc_inv, -fx * lc[1] * zc_inv2};
        double A[6];  // J_pi * Rcw
        for (int r = 0; r < 2; ++r)
            for (int c = 0; c < 3; ++c) A[r * 3 + c] = Jpi[r * 3] * Rcw[c] + Jpi[r * 3 + 1] * Rcw[3 + c] + Jpi[r * 3 + 2] * Rcw[6 + c];
        const double d[3] = {lw[0] - pi[0], lw[1] - pi[1], lw[2] - pi[2]};
        const double sk[9] = {0, -d[2], d[1], d[2], 0, -d[0], -d[1], d[0], 0};
        double Jr[4];  // (A * skew(d))[:, 0:2]
        for (int r = 0; r < 2; ++r)
            for (int c = 0; c < 2; ++c) Jr[r * 2 + c] = A[r * 3] * sk[c] + A[r * 3 + 1] * sk[3 + c] + A[r * 3 + 2] * sk[6 + c];
        const double Jz[2] = {-A[2], -A[5]};
        const float Sigma_rotxy = (float)(1. / xrot_info);
        const float Sigma_z = (float)(1. / z_info);
        const double s2 = sigma2[k];
        double S[4];
        for (int r = 0; r < 2; ++r)
            for (int c = 0; c < 2; ++c)
                S[r * 2 + c] = Sigma_rotxy * (Jr[r * 2] * Jr[c * 2] + Jr[r * 2 + 1] * Jr[c * 2 + 1]) + Sigma_z * Jz[r] * Jz[c] +
                               (r == c ? s2 : 0.0);
        const double det = S[0] * S[3] - S[1] * S[2];
        const double id = 1.0 / det;
        out[4 * k + 0] = S[3] * id; out[4 * k + 1] = -S[1] * id;
        out[4 * k + 2] = -S[2] * id; out[4 * k + 3] = S[0] * id;
    }
}

// mode 0 = Levenberg-Marquardt with g2o's policy (the reference behaviour, optimizer.h:32)
// mode 1 = plain Gauss-Newton (lambda = 0, every step accepted) - "GN iteration" of BASELINE.json
// stop_flag mirrors SparseOptimizer::setForceStopFlag (LocalMapper.cpp:246); may be NULL.
int ba_ref_optimize(const ba_ref_problem* p, int iters, int mode, const volatile uint8_t* stop_flag, double* poses_out,
                    double* lms_out, ba_ref_stats* stats) {
    Cam cam = make_cam(*p);
    State st, trial;
    st.poses.assign(p->poses, p->poses + 3 * p->P);
    st.lms.assign(p->lms, p->lms + 3 * (size_t)p->L);
    System sys;
    build_csr(*p, sys);
    const int n = 3 * p->P;
    ba_ref_stats s;
    std::memset(&s, 0, sizeof(s));
    double lambda = 0, ni = 2;
    std::vector<double> S, bs, Dinv, xl;
    s.chi2_init = robust_chi2(*p, cam, st);
    s.chi2_final = s.chi2_init;
    bool ok = true;
    auto terminate = [&]() { return stop_flag && *stop_flag; };
    for (int it = 0; it < iters && !terminate() && ok; ++it) {
        double currentChi = robust_chi2(*p, cam, st);
        build_system(*p, cam, st, sys);
        if (mode == 0 && it == 0) {  // computeLambdaInit: tau * max |diag(H)| over all free vertices
            double maxd = 0;
            for (int k = 0; k < p->P; ++k)
                if (!p->fixed[k])
                    for (int r = 0; r < 3; ++r) maxd = std::max(std::fabs(sys.Hpp[(size_t)(3 * k + r) * n + 3 * k + r]), maxd);
            for (int l = 0; l < p->L; ++l)
                for (int r = 0; r < 3; ++r) maxd = std::max(std::fabs(sys.Hll[(size_t)l * 9 + r * 4]), maxd);
            lambda = 1e-5 * maxd;
            ni = 2;
        }
        double rho = 0;
        int qmax = 0;
        do {
            schur(*p, sys, mode == 0 ? lambda : 0.0, S, bs, Dinv);
            std::vector<double> Lm = S, xp = bs;
            const bool ok2 = cholesky(Lm, n);
            if (ok2) chol_solve(Lm, n, xp); else std::fill(xp.begin(), xp.end(), 0.0);
            back_substitute(*p, sys, Dinv, xp, xl);
            apply_update(*p, st, xp, xl, trial);
            double tempChi = robust_chi2(*p, cam, trial);
            if (!ok2) tempChi = std::numeric_limits<double>::max();
            ++s.trials;
            ++qmax;
            if (mode == 1) {
                st = trial; currentChi = tempChi; rho = 1;
                break;
            }
            rho = currentChi - tempChi;
            double scale = 0;  // computeScale: sum x_j (lambda x_j + b_j) over the full system
            for (int k = 0; k < p->P; ++k)
                if (!p->fixed[k])
                    for (int r = 0; r < 3; ++r) scale += xp[3 * k + r] * (lambda * xp[3 * k + r] + sys.bp[3 * k + r]);
            for (size_t i = 0; i < xl.size(); ++i) scale += xl[i] * (lambda * xl[i] + sys.bl[i]);
            scale += 1e-3;
            rho /= scale;
            if (s.n_rho < 256) s.rho_log[s.n_rho++] = rho;
            if (rho > 0 && std::isfinite(tempChi)) {
                double alpha = 1. - std::pow(2 * rho - 1, 3);
                alpha = std::min(alpha, 2. / 3.);
                const double scaleFactor = std::max(1. / 3., alpha);
                lambda *= scaleFactor;
                ni = 2;
                currentChi = tempChi;
                st = trial;
            } else {
                lambda *= ni;
                ni *= 2;
            }
        } while (rho < 0 && qmax < 10 && !terminate());
        if (it < 64) { s.chi2_hist[it] = currentChi; s.lambda_hist[it] = lambda; s.trials_hist[it] = qmax; }
        s.iterations = it + 1;
        s.chi2_final = currentChi;
        if (mode == 0 && (qmax == 10 || rho == 0)) { s.terminated = 1; ok = false; }
    }
    s.lambda_final = lambda;
    if (poses_out) std::memcpy(poses_out, st.poses.data(), st.poses.size() * sizeof(double));
    if (lms_out) std::memcpy(lms_out, st.lms.data(), st.lms.size() * sizeof(double));
    if (stats) *stats = s;
    return 0;
}

// ---------------------------------------------------------------------------------------------
// Localizer::DoLocalBA (src/Localizer.cpp:233-302): pose-only bundle adjustment - SURVEY.md 8(f).2.
//   one VertexSE3Expmap (Tcw, free), the observed map points as FIXED non-marginalised VertexSBAPointXYZ,
//   one EdgeProjectXYZ2UV per observation (information invSigma2 * I, Huber delta = Config::TH_HUBER),
//   one EdgeSE3ExpmapPrior from addPlaneMotionSE3Expmap (src/optimizer.cpp:236-314, 159-197), LM, optimize(30).
// [3P g2o 20160424, restated from memory]: types_six_dof_expmap (EdgeProjectXYZ2UV::computeError / linearizeOplus,
// VertexSE3Expmap::oplusImpl = exp(update) * estimate), se3quat.h (exp, log, adj), the Levenberg policy shared with
// ba_ref_optimize above.  Rotations are kept as matrices and passed through a unit quaternion after every product /
// exponential, the way SE3Quat::normalizeRotation does.
// pose12 = rotation row-major (9) then translation (3): x_c = R x_w + t.
// ---------------------------------------------------------------------------------------------
}  // extern "C"

#include "se3_ref.h"

namespace {

struct PoseProblem {
    int n;
    const double *xyz, *uv, *w;
    double f, cx, cy, delta;
    Se3 prior;
    const double* prior_info;   // 6x6 row-major
};

// robust chi2 of all edges at T; when H / b are given also the normal equations (21 upper entries filled symmetric)
double pose_system(const PoseProblem& p, const Se3& T, double* H, double* b) {
    if (H) { std::memset(H, 0, 36 * sizeof(double)); std::memset(b, 0, 6 * sizeof(double)); }
    double chi = 0;
    for (int i = 0; i < p.n; ++i) {
        const double* X = p.xyz + 3 * i;
        const double x = T.R[0] * X[0] + T.R[1] * X[1] + T.R[2] * X[2] + T.t[0];
        const double y = T.R[3] * X[0] + T.R[4] * X[1] + T.R[5] * X[2] + T.t[1];
        const double z = T.R[6] * X[0] + T.R[7] * X[1] + T.R[8] * X[2] + T.t[2];
        const double e0 = p.uv[2 * i] - (x / z * p.f + p.cx), e1 = p.uv[2 * i + 1] - (y / z * p.f + p.cy);
        const double w = p.w[i];
        const double e2 = w * (e0 * e0 + e1 * e1);
        double rho0, rho1;
        if (e2 <= p.delta * p.delta) { rho0 = e2; rho1 = 1; }
        else { const double sq = std::sqrt(e2); rho0 = 2 * sq * p.delta - p.delta * p.delta; rho1 = p.delta / sq; }
        chi += rho0;
        if (!H) continue;
        const double z2 = z * z, f = p.f;
        const double J[2][6] = {{x * y / z2 * f, -(1 + (x * x / z2)) * f, y / z * f, -1. / z * f, 0, x / z2 * f},
                                {(1 + y * y / z2) * f, -x * y / z2 * f, -x / z * f, 0, -1. / z * f, y / z2 * f}};
        const double ww = rho1 * w;
        for (int a = 0; a < 6; ++a) {
            b[a] += -ww * (J[0][a] * e0 + J[1][a] * e1);
            for (int c = 0; c < 6; ++c) H[6 * a + c] += ww * (J[0][a] * J[0][c] + J[1][a] * J[1][c]);
        }
    }
    // EdgeSE3ExpmapPrior: error = log(measurement * estimate^-1), Jacobian = -I, no robust kernel
    double ep[6];
    se3_log(se3_mul(p.prior, se3_inv(T)), ep);
    for (int a = 0; a < 6; ++a) {
        double s = 0;
        for (int c = 0; c < 6; ++c) s += p.prior_info[6 * a + c] * ep[c];
        chi += ep[a] * s;
        if (H) {
            b[a] += s;
            for (int c = 0; c < 6; ++c) H[6 * a + c] += p.prior_info[6 * a + c];
        }
    }
    return chi;
}

bool solve6(const double* H, double lambda, const double* b, double* x) {
    double L[36];
    for (int i = 0; i < 6; ++i)
        for (int j = 0; j < 6; ++j) L[6 * i + j] = H[6 * i + j] + (i == j ? lambda : 0.0);
    for (int j = 0; j < 6; ++j) {
        double d = L[6 * j + j];
        for (int k = 0; k < j; ++k) d -= L[6 * j + k] * L[6 * j + k];
        if (!(d > 0)) return false;
        d = std::sqrt(d);
        L[6 * j + j] = d;
        for (int i = j + 1; i < 6; ++i) {
            double v = L[6 * i + j];
            for (int k = 0; k < j; ++k) v -= L[6 * i + k] * L[6 * j + k];
            L[6 * i + j] = v / d;
        }
    }
    double y[6];
    for (int i = 0; i < 6; ++i) {
        double v = b[i];
        for (int k = 0; k < i; ++k) v -= L[6 * i + k] * y[k];
        y[i] = v / L[6 * i + i];
    }
    for (int i = 5; i >= 0; --i) {
        double v = y[i];
        for (int k = i + 1; k < 6; ++k) v -= L[6 * k + i] * x[k];
        x[i] = v / L[6 * i + i];
    }
    return true;
}

}  // namespace

extern "C" {

// addPlaneMotionSE3Expmap (src/optimizer.cpp:236-314, the non-Euler branch): measurement = the pose with roll, pitch and
// height of the BODY removed, information = adj(Tbc)^T diag(xrot, yrot, 1e-4, 1e-4, 1e-4, z) adj(Tbc), made symmetric
// by copying the upper triangle (:296-298)
void ba_ref_plane_motion_prior(const double* Tcw12, const double* Tbc12, double xrot_info, double yrot_info, double z_info,
                               double* meas12, double* info36) {
    const Se3 Tcw = se3_from(Tcw12), Tbc = se3_from(Tbc12);
    Se3 Tbw = se3_mul(Tbc, Tcw);
    const double yaw = rotation_vector_z(Tbw.R);   // z of the rotation vector of Rbw, Eigen::AngleAxisd(quaternion): angle * axis
    const double c = std::cos(yaw), sn = std::sin(yaw);
    const double Rz[9] = {c, -sn, 0, sn, c, 0, 0, 0, 1};
    std::memcpy(Tbw.R, Rz, sizeof(Rz));
    Tbw.t[2] = 0;
    se3_to(se3_mul(se3_inv(Tbc), Tbw), meas12);
    double A[36] = {0}, sk[9], sR[9];         // SE3Quat::adj: [R 0; skew(t) R  R], vector order (rot, trans)
    skew3(Tbc.t, sk);
    mat3_mul(sk, Tbc.R, sR);
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            A[6 * i + j] = Tbc.R[3 * i + j];
            A[6 * (i + 3) + (j + 3)] = Tbc.R[3 * i + j];
            A[6 * (i + 3) + j] = sR[3 * i + j];
        }
    const double D[6] = {xrot_info, yrot_info, 1e-4, 1e-4, 1e-4, z_info};
    for (int i = 0; i < 6; ++i)
        for (int j = 0; j < 6; ++j) {
            double s = 0;
            for (int k = 0; k < 6; ++k) s += A[6 * k + i] * D[k] * A[6 * k + j];
            info36[6 * i + j] = s;
        }
    for (int i = 0; i < 6; ++i)
        for (int j = 0; j < i; ++j) info36[6 * i + j] = info36[6 * j + i];
}

int ba_ref_pose_only(const double* Tcw12, const double* prior_meas12, const double* prior_info36, int n, const double* xyz,
                     const double* uv, const double* inv_sigma2, double f, double cx, double cy, double delta, int iters,
                     double* Tcw_out12, ba_ref_stats* stats) {
    PoseProblem p{n, xyz, uv, inv_sigma2, f, cx, cy, delta, se3_from(prior_meas12), prior_info36};
    Se3 est = se3_from(Tcw12);
    ba_ref_stats s;
    std::memset(&s, 0, sizeof(s));
    double lambda = 0, ni = 2, H[36], b[6], x[6];
    s.chi2_init = pose_system(p, est, nullptr, nullptr);
    s.chi2_final = s.chi2_init;
    bool ok = true;
    for (int it = 0; it < iters && ok; ++it) {
        double currentChi = pose_system(p, est, H, b);
        if (it == 0) {
            double maxd = 0;
            for (int r = 0; r < 6; ++r) maxd = std::max(std::fabs(H[7 * r]), maxd);
            lambda = 1e-5 * maxd;
            ni = 2;
        }
        double rho = 0;
        int qmax = 0;
        do {
            const bool ok2 = solve6(H, lambda, b, x);
            if (!ok2) std::memset(x, 0, sizeof(x));
            const Se3 trial = se3_mul(se3_exp(x), est);
            double tempChi = pose_system(p, trial, nullptr, nullptr);
            if (!ok2) tempChi = std::numeric_limits<double>::max();
            ++s.trials;
            ++qmax;
            rho = currentChi - tempChi;
            double scale = 1e-3;
            for (int r = 0; r < 6; ++r) scale += x[r] * (lambda * x[r] + b[r]);
            rho /= scale;
            if (rho > 0 && std::isfinite(tempChi)) {
                double alpha = 1. - std::pow(2 * rho - 1, 3);
                alpha = std::min(alpha, 2. / 3.);
                lambda *= std::max(1. / 3., alpha);
                ni = 2;
                currentChi = tempChi;
                est = trial;
            } else {
                lambda *= ni;
                ni *= 2;
            }
        } while (rho < 0 && qmax < 10);
        if (it < 64) { s.chi2_hist[it] = currentChi; s.lambda_hist[it] = lambda; s.trials_hist[it] = qmax; }
        s.iterations = it + 1;
        s.chi2_final = currentChi;
        if (qmax == 10 || rho == 0) { s.terminated = 1; ok = false; }
    }
    s.lambda_final = lambda;
    se3_to(est, Tcw_out12);
    if (stats) *stats = s;
    return 0;
}

// test hooks: the building blocks of ba_ref_pose_only, so that tests can check them against their definitions
// (exp / log round trips, analytic Jacobian of EdgeProjectXYZ2UV against a numeric derivative through exp)
void ba_ref_se3_exp(const double* update6, double* pose12) { se3_to(se3_exp(update6), pose12); }
void ba_ref_se3_log(const double* pose12, double* out6) { se3_log(se3_from(pose12), out6); }
void ba_ref_se3_mul(const double* a12, const double* b12, double* out12) { se3_to(se3_mul(se3_from(a12), se3_from(b12)), out12); }
// error (2) and the 2x6 Jacobian w.r.t. the pose update (rotation, translation) of one EdgeProjectXYZ2UV
void ba_ref_project_edge(const double* pose12, const double* X, const double* uv, double f, double cx, double cy,
                         double* e2, double* J12) {
    const Se3 T = se3_from(pose12);
    const double x = T.R[0] * X[0] + T.R[1] * X[1] + T.R[2] * X[2] + T.t[0];
    const double y = T.R[3] * X[0] + T.R[4] * X[1] + T.R[5] * X[2] + T.t[1];
    const double z = T.R[6] * X[0] + T.R[7] * X[1] + T.R[8] * X[2] + T.t[2];
    e2[0] = uv[0] - (x / z * f + cx);
    e2[1] = uv[1] - (y / z * f + cy);
    const double z2 = z * z;
    const double J[12] = {x * y / z2 * f, -(1 + (x * x / z2)) * f, y / z * f, -1. / z * f, 0, x / z2 * f,
                          (1 + y * y / z2) * f, -x * y / z2 * f, -x / z * f, 0, -1. / z * f, y / z2 * f};
    std::memcpy(J12, J, sizeof(J));
}

}  // extern "C"
