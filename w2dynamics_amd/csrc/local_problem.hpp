// Host-side construction of the local impurity problem in block form - what init_solver does before the sampler
// starts (CTQMC.F90:7957-8031): Fock states, blocks from the QuantumNumbers string (States.F90:149-184, 98-144),
// creation/annihilation operators and Connect (Operator.F90:115-245), H_loc = one-particle part + U
// (482-542, 1042-1125), its eigen-decomposition per block (637-670, 754-801) and the operators in the eigenbasis
// (1130-1168).  The eigen-solver is a cyclic Jacobi iteration (blocks are small); eigenvectors are only defined up
// to sign / rotations inside degenerate subspaces, which leaves every trace and every basis-independent observable
// unchanged.
#pragma once
#include <algorithm>
#include <cmath>
#include <string>
#include <vector>

namespace w2dyn {

struct LocalProblem {
    int nd = 0, nflav = 0, nstates = 0, nsst = 0, nsmax = 0;
    std::vector<int> sst_of_state, sst_n, sst_off, states, idx_in_block;   // idx_in_block[Fock state] = position in its block
    std::vector<double> eval;                       // block order, ascending inside a block
    std::vector<std::vector<double>> evec;          // per block, column major n x n: evec[b][j + n*i] = <Fock j|eigen i>
    std::vector<int> conn;                          // [(sst*nflav + f)*2 + (ca-1)]
    std::vector<int> shift;                         // same index: block whose quantum numbers are those of `sst` changed by the
                                                    // operator (formal arithmetic, whether or not a state of sst admits it); -1: none
    std::vector<std::vector<double>> op;            // same index: n_target x n_source column major (empty if not connected)
    double egs = 0.0;
    bool identity_blocks = false;                   // every block holds exactly its own Fock state (index == state)

    int bit(int f) const { return (f >> 1) + nd * (f & 1); }
    static double fsign(int state, int b) { return (__builtin_popcount((unsigned)state & ((1u << b) - 1u)) & 1) ? -1.0 : 1.0; }
    int connect(int sst, int f, int ca) const { return conn[((size_t)sst * nflav + f) * 2 + (ca - 1)]; }
    const std::vector<double> &oper(int sst, int f, int ca) const { return op[((size_t)sst * nflav + f) * 2 + (ca - 1)]; }

    // quantum numbers of a Fock state (States.F90:188-325)
    double qn(int which, int s) const
    {
        double v = 0.0;
        switch (which) {
        case 0: return (double)__builtin_popcount((unsigned)s);                                                     // Nt
        case 1: for (int b = 0; b < nd; ++b) { if (s >> b & 1) v -= 0.5; if (s >> (b + nd) & 1) v += 0.5; } return v;   // Szt
        case 2: { double p = 1.0; for (int b = 0; b < nd; ++b, p *= 10.0) if ((s >> b & 1) != (s >> (b + nd) & 1)) v += p; return v; }   // Qzt
        case 3: { double p = 1.0; for (int b = 0; b < 2 * nd; ++b, p *= 10.0) if (s >> b & 1) v += p; return v; }  // Azt
        case 4: for (int sp = 0; sp < 2; ++sp) for (int b = 1; b <= nd; ++b) if (s >> ((b - 1) + nd * sp) & 1) v += (double)b - (double)(nd + 1) / 2.0; return v;   // Lzt
        default: return qn(4, s) + qn(1, s);                                                                        // Jzt
        }
    }

    static void jacobi(std::vector<double> &A, int n, double *w, std::vector<double> &V)
    {
        V.assign((size_t)n * n, 0.0);
        for (int i = 0; i < n; ++i) V[i + (size_t)n * i] = 1.0;
        for (int sweep = 0; sweep < 100; ++sweep) {
            double off = 0.0;
            for (int p = 0; p < n; ++p) for (int q = p + 1; q < n; ++q) off += A[p * n + q] * A[p * n + q];
            if (off < 1e-300) break;
            for (int p = 0; p < n; ++p) for (int q = p + 1; q < n; ++q) {
                const double apq = A[p * n + q];
                if (apq == 0.0) continue;
                const double theta = (A[q * n + q] - A[p * n + p]) / (2.0 * apq);
                const double t = (theta >= 0.0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
                const double c = 1.0 / std::sqrt(t * t + 1.0), sn = t * c;
                for (int k = 0; k < n; ++k) { double x = A[k * n + p], y = A[k * n + q]; A[k * n + p] = c * x - sn * y; A[k * n + q] = sn * x + c * y; }
                for (int k = 0; k < n; ++k) { double x = A[p * n + k], y = A[q * n + k]; A[p * n + k] = c * x - sn * y; A[q * n + k] = sn * x + c * y; }
                for (int k = 0; k < n; ++k) { double x = V[k + (size_t)n * p], y = V[k + (size_t)n * q]; V[k + (size_t)n * p] = c * x - sn * y; V[k + (size_t)n * q] = sn * x + c * y; }
            }
        }
        for (int i = 0; i < n; ++i) w[i] = A[i * n + i];
        for (int i = 0; i < n; ++i) {
            int m = i;
            for (int j = i + 1; j < n; ++j) if (w[j] < w[m]) m = j;
            if (m != i) { std::swap(w[i], w[m]); for (int k = 0; k < n; ++k) std::swap(V[k + (size_t)n * i], V[k + (size_t)n * m]); }
        }
    }

    // matrix elements of H_loc = one-particle part + U between Fock states (Operator.F90:482-542, 1042-1107), handed to
    // `add(source state, target state, value)`.  The reference multiplies block matrices along the chain of `Connect`
    // entries, so a term only counts if every single operator of it leads from a block into the block `Connect` names
    // (never into the same block, Operator.F90:150): `step(state before, state after, flavour, ca)` decides that.
    template <class F, class S> void for_each_h_element(const double *U, const double *mu_full, F add, S step) const
    {
        const int nf = nflav, ns = nstates;
        auto annihilate = [&](int &st, double &sg, int f) { int b = bit(f); if (!(st >> b & 1)) return false; sg *= fsign(st, b); const int t = st ^ (1 << b); if (!step(st, t, f, 2)) return false; st = t; return true; };
        auto create = [&](int &st, double &sg, int f) { int b = bit(f); if (st >> b & 1) return false; sg *= fsign(st, b); const int t = st ^ (1 << b); if (!step(st, t, f, 1)) return false; st = t; return true; };
        for (int f1 = 0; f1 < nf; ++f1) for (int f2 = 0; f2 < nf; ++f2) {
            double mu = mu_full[((size_t)((f1 >> 1) * 2 + (f1 & 1)) * nd + (f2 >> 1)) * 2 + (f2 & 1)];
            if (mu == 0.0) continue;
            for (int s = 0; s < ns; ++s) { int t = s; double sg = 1.0; if (annihilate(t, sg, f2) && create(t, sg, f1)) add(s, t, mu * sg); }
        }
        for (int f1 = 0; f1 < nf; ++f1) for (int f2 = 0; f2 < nf; ++f2) for (int f3 = 0; f3 < nf; ++f3) for (int f4 = 0; f4 < nf; ++f4) {
            double u = 0.5 * U[(((size_t)f1 * nf + f2) * nf + f3) * nf + f4];
            if (u == 0.0) continue;
            for (int s = 0; s < ns; ++s) {
                int t = s; double sg = 1.0;
                if (annihilate(t, sg, f3) && annihilate(t, sg, f4) && create(t, sg, f2) && create(t, sg, f1)) add(s, t, u * sg);
            }
        }
    }

    // QuantumNumbers = All: analyze_hamiltonian (Operator.F90:303-442).  Starting from the partition the named quantum
    // numbers give (one block if none is named), (i) every block is split into the connected components of H_loc
    // (elements above `threshold`), numbered block by block by their lowest Fock state, (ii) blocks that one creation or
    // annihilation operator reaches from the same block are joined until every operator maps a block into one block,
    // (iii) the joined blocks keep the order of their lowest original number.  Refines sst_of_state / nsst in place.
    void analyze_hamiltonian(const double *U, const double *mu_full, double threshold)
    {
        const int nf = nflav, ns = nstates;
        std::vector<double> H((size_t)ns * ns, 0.0);
        // Connect of the starting partition: the last block found that the operator reaches, none for the block itself
        std::vector<int> conn0((size_t)nsst * nf * 2, -1);
        for (int b = 0; b < nsst; ++b) for (int tb = 0; tb < nsst; ++tb) {
            if (tb == b) continue;
            for (int s = 0; s < ns; ++s) {
                if (sst_of_state[s] != b) continue;
                for (int f = 0; f < nf; ++f) { const int t = s ^ (1 << bit(f)); if (sst_of_state[t] == tb) conn0[((size_t)b * nf + f) * 2 + (t > s ? 0 : 1)] = tb; }
            }
        }
        for_each_h_element(U, mu_full, [&](int src, int dst, double v) { if (sst_of_state[src] == sst_of_state[dst]) H[(size_t)dst * ns + src] += v; },
                           [&](int cur, int next, int f, int ca) { return conn0[((size_t)sst_of_state[cur] * nf + f) * 2 + (ca - 1)] == sst_of_state[next]; });
        for (int i = 0; i < ns; ++i) for (int j = i + 1; j < ns; ++j) { double v = (H[(size_t)i * ns + j] + H[(size_t)j * ns + i]) / 2.0; H[(size_t)i * ns + j] = H[(size_t)j * ns + i] = v; }
        std::vector<int> comp(ns, -1), stack;
        int ncomp = 0;
        for (int b = 0; b < nsst; ++b) for (int s0 = 0; s0 < ns; ++s0) {
            if (sst_of_state[s0] != b || comp[s0] != -1) continue;
            comp[s0] = ncomp; stack.assign(1, s0);
            while (!stack.empty()) {
                const int a = stack.back(); stack.pop_back();
                for (int t = 0; t < ns; ++t) {
                    const double h = H[(size_t)a * ns + t];
                    if (h != 0.0 && std::fabs(h) > threshold && sst_of_state[t] == b && comp[t] == -1) { comp[t] = ncomp; stack.push_back(t); }
                }
            }
            ++ncomp;
        }
        std::vector<int> parent(ncomp);
        for (int c = 0; c < ncomp; ++c) parent[c] = c;
        auto find = [&](int c) { while (parent[c] != c) c = parent[c] = parent[parent[c]]; return c; };
        for (bool change = true; change;) {
            change = false;
            for (int f = 0; f < nf; ++f) for (int ca = 1; ca <= 2; ++ca) {
                std::vector<int> target(ncomp, -1);
                for (int s = 0; s < ns; ++s) {
                    if ((ca == 1) == (bool)(s >> bit(f) & 1)) continue;
                    const int c = find(comp[s]), t = find(comp[s ^ (1 << bit(f))]);
                    if (target[c] == -1) target[c] = t;
                    else if (find(target[c]) != t) { const int r = find(target[c]); parent[std::max(r, t)] = std::min(r, t); target[c] = std::min(r, t); change = true; }
                }
            }
        }
        std::vector<int> label(ncomp, -1);
        nsst = 0;
        for (int c = 0; c < ncomp; ++c) if (find(c) == c) label[c] = nsst++;      // roots are the lowest original numbers
        for (int s = 0; s < ns; ++s) sst_of_state[s] = label[find(comp[s])];
    }

    // U: (2Nd)^4 with flavour 2*orb+spin; mu_full: (Nd,2,Nd,2) as passed to init_solver; qns: QuantumNumbers string;
    // epsblock: EPSBLOCK (only used with QuantumNumbers = All, CTQMC.F90:7971-7975)
    void build(int nd_, const double *U, const double *mu_full, std::string qns, double epsevec, double epsblock = 1e-13)
    {
        nd = nd_; nflav = 2 * nd; nstates = 1 << nflav;
        const int nf = nflav, ns = nstates;
        const bool automatic = qns.find("All") != std::string::npos;
        const char *names[6] = {"Nt", "Szt", "Qzt", "Azt", "Lzt", "Jzt"};
        bool use[6];
        for (int q = 0; q < 6; ++q) use[q] = qns.find(names[q]) != std::string::npos;
        sst_of_state.assign(ns, -1);
        nsst = 0;
        for (int s1 = 0; s1 < ns; ++s1) {
            if (sst_of_state[s1] != -1) continue;
            sst_of_state[s1] = nsst++;
            for (int s2 = s1; s2 < ns; ++s2) {
                if (sst_of_state[s2] != -1) continue;
                bool same = true;
                for (int q = 0; q < 6 && same; ++q) if (use[q] && !(std::fabs(qn(q, s1) - qn(q, s2)) < 1e-12)) same = false;
                if (same) sst_of_state[s2] = nsst - 1;
            }
        }
        if (automatic) {
            double big = 0.0;
            for (size_t i = 0; i < (size_t)nf * nf; ++i) big = std::max(big, std::fabs(mu_full[i]));
            for (size_t i = 0; i < (size_t)nf * nf * nf * nf; ++i) big = std::max(big, std::fabs(U[i]));
            analyze_hamiltonian(U, mu_full, epsblock * big);
        }
        sst_n.assign(nsst, 0); sst_off.assign(nsst, 0); states.assign(ns, 0); idx_in_block.assign(ns, 0);
        for (int s = 0; s < ns; ++s) sst_n[sst_of_state[s]] += 1;
        for (int b = 1; b < nsst; ++b) sst_off[b] = sst_off[b - 1] + sst_n[b - 1];
        {
            std::vector<int> fill(nsst, 0);
            for (int s = 0; s < ns; ++s) { int b = sst_of_state[s]; idx_in_block[s] = fill[b]; states[sst_off[b] + fill[b]++] = s; }
        }
        nsmax = *std::max_element(sst_n.begin(), sst_n.end());
        identity_blocks = nsmax == 1;
        for (int s = 0; s < ns && identity_blocks; ++s) if (sst_of_state[s] != s) identity_blocks = false;

        conn.assign((size_t)nsst * nf * 2, -1);
        for (int s = 0; s < ns; ++s) for (int f = 0; f < nf; ++f) {
            int t = s ^ (1 << bit(f)), b = sst_of_state[s], tb = sst_of_state[t];
            if (tb != b) conn[((size_t)b * nf + f) * 2 + (t > s ? 0 : 1)] = tb;
        }

        // Blocks are in one-to-one correspondence with tuples of the (additive) quantum numbers in use, so the block a
        // sequence passes through after one more operator earlier in time follows from quantum-number arithmetic alone.
        // The sampler uses this table to propose the whole block sequence of a trial configuration at once and then only
        // verifies every link against `conn` (check_sst_sequence, Trace.F90:1243-1302, without its serial walk).
        shift.assign((size_t)nsst * nf * 2, -1);
        if (automatic) shift = conn;      // after analyze_hamiltonian every operator maps a block into exactly one block (or none)
        else {
            std::vector<std::vector<double>> qb(nsst, std::vector<double>(6, 0.0));
            for (int b = 0; b < nsst; ++b) for (int q = 0; q < 6; ++q) qb[b][q] = qn(q, states[sst_off[b]]);
            for (int b = 0; b < nsst; ++b) for (int f = 0; f < nf; ++f) for (int ca = 1; ca <= 2; ++ca) {
                const double sg = ca == 1 ? 1.0 : -1.0;
                const int orb = f >> 1, sp = f & 1, bt = bit(f);
                double t[6];
                t[0] = qb[b][0] + sg;
                t[1] = qb[b][1] + sg * (sp ? 0.5 : -0.5);
                { const long long v = std::llround(qb[b][2]); const long long p10 = std::llround(std::pow(10.0, orb)); const int dig = (int)((v / p10) % 10); t[2] = qb[b][2] + (dig ? -(double)p10 : (double)p10); }
                { const long long v = std::llround(qb[b][3]); const long long p10 = std::llround(std::pow(10.0, bt)); const int dig = (int)((v / p10) % 10);
                  t[3] = ((ca == 1 && dig) || (ca == 2 && !dig)) ? -1.0 : qb[b][3] + sg * (double)p10; }
                const double dl = (double)(orb + 1) - (double)(nd + 1) / 2.0;
                t[4] = qb[b][4] + sg * dl;
                t[5] = t[4] + t[1];
                if (use[3] && t[3] < 0.0) continue;
                for (int c = 0; c < nsst; ++c) {
                    bool same = true;
                    for (int q = 0; q < 6 && same; ++q) if (use[q] && !(std::fabs(qb[c][q] - t[q]) < 1e-9)) same = false;
                    if (same) { shift[((size_t)b * nf + f) * 2 + (ca - 1)] = c; break; }
                }
            }
        }

        std::vector<std::vector<double>> H(nsst);
        for (int b = 0; b < nsst; ++b) H[b].assign((size_t)sst_n[b] * sst_n[b], 0.0);
        auto add = [&](int src, int dst, double v) {
            int b = sst_of_state[src];
            if (sst_of_state[dst] != b) return;
            H[b][(size_t)idx_in_block[dst] * sst_n[b] + idx_in_block[src]] += v;
        };
        for_each_h_element(U, mu_full, add, [](int, int, int, int) { return true; });
        eval.assign(ns, 0.0); evec.assign(nsst, {});
        for (int b = 0; b < nsst; ++b) {
            const int n = sst_n[b];
            for (int i = 0; i < n; ++i) for (int j = i + 1; j < n; ++j) { double v = (H[b][i * n + j] + H[b][j * n + i]) / 2.0; H[b][i * n + j] = H[b][j * n + i] = v; }
            jacobi(H[b], n, &eval[sst_off[b]], evec[b]);
            for (int i = 0; i < n; ++i) {                              // EPSEVEC and the reference's renormalisation
                double sum = 0.0;
                for (int j = 0; j < n; ++j) { double &v = evec[b][j + (size_t)n * i]; if (std::fabs(v) < epsevec) v = 0.0; sum += v * v; }
                for (int j = 0; j < n; ++j) evec[b][j + (size_t)n * i] /= sum;
            }
        }
        egs = eval[0];
        for (int b = 0; b < nsst; ++b) egs = std::min(egs, eval[sst_off[b]]);

        op.assign((size_t)nsst * nf * 2, {});
        for (int b = 0; b < nsst; ++b) for (int f = 0; f < nf; ++f) for (int ca = 1; ca <= 2; ++ca) {
            const int tb = connect(b, f, ca);
            if (tb < 0) continue;
            const int n1 = sst_n[b], n2 = sst_n[tb], bt = bit(f);
            std::vector<double> P((size_t)n2 * n1, 0.0), T((size_t)n2 * n1, 0.0), O((size_t)n2 * n1, 0.0);
            for (int i1 = 0; i1 < n1; ++i1) {
                int s = states[sst_off[b] + i1];
                if ((ca == 1) == (bool)(s >> bt & 1)) continue;
                int t = s ^ (1 << bt);
                if (sst_of_state[t] == tb) P[idx_in_block[t] + (size_t)n2 * i1] = fsign(s, bt);
            }
            for (int i = 0; i < n2; ++i) for (int j = 0; j < n1; ++j) { double a = 0.0; for (int k = 0; k < n1; ++k) a += P[i + (size_t)n2 * k] * evec[b][k + (size_t)n1 * j]; T[i + (size_t)n2 * j] = a; }
            for (int i = 0; i < n2; ++i) for (int j = 0; j < n1; ++j) { double a = 0.0; for (int k = 0; k < n2; ++k) a += evec[tb][k + (size_t)n2 * i] * T[k + (size_t)n2 * j]; O[i + (size_t)n2 * j] = a; }
            op[((size_t)b * nf + f) * 2 + (ca - 1)] = std::move(O);
        }
    }

    // S^2 = 1/2 (S+S- + S-S+) + Sz^2 in the occupation basis, block by block (init_S2alt2, Operator.F90:548-633: strings
    // c+_{b3 s4} c_{b3 s3} c+_{b1 s2} c_{b1 s1} with the coefficients sm1(s2,s1) sm2(s4,s3) of the three terms), column major
    // n x n per block.  Returns false when a string leaves the block it started from: the reference then adds the element
    // to a position of the wrong block (or outside it), so there is nothing defined to reproduce.
    std::vector<std::vector<double>> s2_occ;
    bool build_s2()
    {
        s2_occ.assign(nsst, {});
        bool closed = true;
        auto apply = [&](int &st, double &sg, int f, int ca) {
            const int b = bit(f);
            if ((ca == 1) == (bool)(st >> b & 1)) return false;
            sg *= fsign(st, b); st ^= 1 << b;
            return true;
        };
        for (int blk = 0; blk < nsst; ++blk) {
            const int n = sst_n[blk];
            s2_occ[blk].assign((size_t)n * n, 0.0);
            for (int i1 = 0; i1 < n; ++i1) for (int xyz = 0; xyz < 3; ++xyz)
                for (int b1 = 0; b1 < nd; ++b1) for (int s1 = 0; s1 < 2; ++s1) for (int s2 = 0; s2 < 2; ++s2) {
                    const double c1 = xyz == 0 ? ((s2 == 1 && s1 == 0) ? 1.0 : 0.0) : (xyz == 1 ? ((s2 == 0 && s1 == 1) ? 1.0 : 0.0) : (s1 == s2 ? (s1 == 0 ? 1.0 : -1.0) : 0.0));
                    if (c1 == 0.0) continue;
                    for (int b3 = 0; b3 < nd; ++b3) for (int s3 = 0; s3 < 2; ++s3) for (int s4 = 0; s4 < 2; ++s4) {
                        const double c2 = xyz == 0 ? ((s4 == 0 && s3 == 1) ? 0.5 : 0.0) : (xyz == 1 ? ((s4 == 1 && s3 == 0) ? 0.5 : 0.0) : (s3 == s4 ? (s3 == 0 ? 0.25 : -0.25) : 0.0));
                        if (c2 == 0.0) continue;
                        int st = states[sst_off[blk] + i1]; double sg = 1.0;
                        if (!apply(st, sg, 2 * b1 + s1, 2) || !apply(st, sg, 2 * b1 + s2, 1) || !apply(st, sg, 2 * b3 + s3, 2) || !apply(st, sg, 2 * b3 + s4, 1)) continue;
                        if (sst_of_state[st] != blk) { closed = false; continue; }
                        s2_occ[blk][idx_in_block[st] + (size_t)n * i1] += c1 * c2 * sg;
                    }
                }
        }
        return closed;
    }

    // <A> = sum_b tr(rho_b A_bb), A = product of operators applied right to left: (fl[0],ca[0]) acts first.
    // rho: block-diagonal density matrix in the eigenbasis, block b at dmoff[b], [bra][ket] row major (CTQMC.F90:2842-3060)
    double expect(const double *rho, const std::vector<int> &dmoff, int nop, const int *fl, const int *ca) const
    {
        double tot = 0.0;
        std::vector<double> v(nsmax), out(nsmax);
        for (int b = 0; b < nsst; ++b) {
            const int n = sst_n[b];
            for (int col = 0; col < n; ++col) {
                std::fill(v.begin(), v.end(), 0.0); v[col] = 1.0;
                int s = b; bool ok = true;
                for (int q = 0; q < nop; ++q) {
                    int t = connect(s, fl[q], ca[q]);
                    if (t < 0) { ok = false; break; }
                    const int n1 = sst_n[s], n2 = sst_n[t]; const std::vector<double> &O = oper(s, fl[q], ca[q]);
                    for (int i = 0; i < n2; ++i) { double a = 0.0; for (int j = 0; j < n1; ++j) a += O[i + (size_t)n2 * j] * v[j]; out[i] = a; }
                    for (int i = 0; i < n2; ++i) v[i] = out[i];
                    s = t;
                }
                if (!ok || s != b) continue;
                for (int row = 0; row < n; ++row) tot += rho[dmoff[b] + col * n + row] * v[row];
            }
        }
        return tot;
    }
};

}  // namespace w2dyn
