/* TEST INFRASTRUCTURE ONLY (oracle) - see locprob.h.  References: /root/reference/src/ctqmc_fortran. */
#include "locprob.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ---- quantum numbers of a Fock state (bit (orb-1) + Nd*(spin-1), spin 1 = down: States.F90:188-325) ---- */
static double qn_nt(int nd, int s) { return (double)__builtin_popcount((unsigned)s & ((1u << (2 * nd)) - 1u)); }
static double qn_szt(int nd, int s)
{
    double v = 0.0;
    for (int b = 0; b < nd; ++b) { if (s >> b & 1) v -= 0.5; if (s >> (b + nd) & 1) v += 0.5; }
    return v;
}
static double qn_qzt(int nd, int s)
{   /* States.F90:219-231: 10**iB for every singly occupied orbital */
    double v = 0.0, p = 1.0;
    for (int b = 0; b < nd; ++b, p *= 10.0) if ((s >> b & 1) != (s >> (b + nd) & 1)) v += p;
    return v;
}
static double qn_azt(int nd, int s)
{   /* States.F90:234-246 */
    double v = 0.0, p = 1.0;
    for (int b = 0; b < 2 * nd; ++b, p *= 10.0) if (s >> b & 1) v += p;
    return v;
}
static double qn_lzt(int nd, int s)
{   /* States.F90:303-316 */
    double v = 0.0;
    for (int sp = 0; sp < 2; ++sp) for (int b = 1; b <= nd; ++b)
        if (s >> ((b - 1) + nd * sp) & 1) v += (double)b - (double)(nd + 1) / 2.0;
    return v;
}
static int qeq(double a, double b) { return fabs(a - b) < 1e-12; }   /* Parameters.F90:48-53 */

static double fermi_sign(int state, int bit) { return (__builtin_popcount((unsigned)state & ((1u << bit) - 1u)) & 1) ? -1.0 : 1.0; }

/* cyclic Jacobi for a symmetric n x n matrix (row major A, destroyed); eigenvalues ascending in w, eigenvectors in
 * the columns of V (column major V[j + n*i]).  Stands in for DSYEV (Operator.F90:637-670). */
static void sym_eig(double *A, int n, double *w, double *V)
{
    for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) V[j + n * i] = (i == j) ? 1.0 : 0.0;
    for (int sweep = 0; sweep < 100; ++sweep) {
        double off = 0.0;
        for (int p = 0; p < n; ++p) for (int q = p + 1; q < n; ++q) off += A[p * n + q] * A[p * n + q];
        if (off < 1e-300) break;
        for (int p = 0; p < n; ++p) for (int q = p + 1; q < n; ++q) {
            double apq = A[p * n + q];
            if (apq == 0.0) continue;
            double theta = (A[q * n + q] - A[p * n + p]) / (2.0 * apq);
            double t = (theta >= 0.0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
            double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
            for (int k = 0; k < n; ++k) {
                double akp = A[k * n + p], akq = A[k * n + q];
                A[k * n + p] = c * akp - s * akq; A[k * n + q] = s * akp + c * akq;
            }
            for (int k = 0; k < n; ++k) {
                double apk = A[p * n + k], aqk = A[q * n + k];
                A[p * n + k] = c * apk - s * aqk; A[q * n + k] = s * apk + c * aqk;
            }
            for (int k = 0; k < n; ++k) {
                double vkp = V[k + n * p], vkq = V[k + n * q];
                V[k + n * p] = c * vkp - s * vkq; V[k + n * q] = s * vkp + c * vkq;
            }
        }
    }
    for (int i = 0; i < n; ++i) w[i] = A[i * n + i];
    for (int i = 0; i < n; ++i) {            /* ascending order */
        int m = i;
        for (int j = i + 1; j < n; ++j) if (w[j] < w[m]) m = j;
        if (m != i) {
            double t = w[i]; w[i] = w[m]; w[m] = t;
            for (int k = 0; k < n; ++k) { double u = V[k + n * i]; V[k + n * i] = V[k + n * m]; V[k + n * m] = u; }
        }
    }
}

#define BIT(fl) (((fl) >> 1) + nd * ((fl) & 1))
static double fermi_sign(int state, int bit);

/* dense H_loc in the Fock basis restricted to the blocks of `part` (oneparticle_hamiltonian 482-542, u_hamiltonian
 * 1042-1107, force_hermitian 1110-1125); H[t * ns + s] = <t|H|s> */
static double *full_hamiltonian(int nd, const double *U, const double *mu_full, const int *part, int nsst)
{
    const int nf = 2 * nd, ns = 1 << nf;
    double *H = (double *)calloc((size_t)ns * ns, sizeof(double));
    /* Connect of this partition (init_Psi, Operator.F90:146-178): the last block iSSt2 /= iSSt1 holding a state that the
     * operator reaches from a state of iSSt1; the Hamiltonian is a sum of block-matrix products along chains of Connect
     * entries (517-534, 1071-1092), so a term only counts if every step follows Connect */
    int *conn0 = (int *)malloc(sizeof(int) * (size_t)nsst * nf * 2);
    for (size_t i = 0; i < (size_t)nsst * nf * 2; ++i) conn0[i] = -1;
    for (int b = 0; b < nsst; ++b) for (int tb = 0; tb < nsst; ++tb) {
        if (tb == b) continue;
        for (int s = 0; s < ns; ++s) {
            if (part[s] != b) continue;
            for (int f = 0; f < nf; ++f) { int t = s ^ (1 << BIT(f)); if (part[t] == tb) conn0[(b * nf + f) * 2 + (t > s ? 0 : 1)] = tb; }
        }
    }
#define STEP(cur, next, f, ca) (conn0[(part[cur] * nf + (f)) * 2 + ((ca) - 1)] == part[next])
    for (int f1 = 0; f1 < nf; ++f1) for (int f2 = 0; f2 < nf; ++f2) {
        double mu = mu_full[(((f1 >> 1) * 2 + (f1 & 1)) * nd + (f2 >> 1)) * 2 + (f2 & 1)];
        if (mu == 0.0) continue;
        for (int s = 0; s < ns; ++s) {
            if (!(s >> BIT(f2) & 1)) continue;
            double sg = fermi_sign(s, BIT(f2)); int t = s ^ (1 << BIT(f2));
            if (!STEP(s, t, f2, 2)) continue;
            if (t >> BIT(f1) & 1) continue;
            sg *= fermi_sign(t, BIT(f1));
            if (!STEP(t, t ^ (1 << BIT(f1)), f1, 1)) continue;
            t ^= 1 << BIT(f1);
            if (part[t] == part[s]) H[(size_t)t * ns + s] += mu * sg;
        }
    }
    for (int f1 = 0; f1 < nf; ++f1) for (int f2 = 0; f2 < nf; ++f2) for (int f3 = 0; f3 < nf; ++f3) for (int f4 = 0; f4 < nf; ++f4) {
        double u = 0.5 * U[((f1 * nf + f2) * nf + f3) * nf + f4];
        if (u == 0.0) continue;
        for (int s = 0; s < ns; ++s) {
            int t = s; double sg = 1.0;
            if (!(t >> BIT(f3) & 1) || !STEP(t, t ^ (1 << BIT(f3)), f3, 2)) continue;
            sg *= fermi_sign(t, BIT(f3)); t ^= 1 << BIT(f3);
            if (!(t >> BIT(f4) & 1) || !STEP(t, t ^ (1 << BIT(f4)), f4, 2)) continue;
            sg *= fermi_sign(t, BIT(f4)); t ^= 1 << BIT(f4);
            if ((t >> BIT(f2) & 1) || !STEP(t, t ^ (1 << BIT(f2)), f2, 1)) continue;
            sg *= fermi_sign(t, BIT(f2)); t ^= 1 << BIT(f2);
            if ((t >> BIT(f1) & 1) || !STEP(t, t ^ (1 << BIT(f1)), f1, 1)) continue;
            sg *= fermi_sign(t, BIT(f1)); t ^= 1 << BIT(f1);
            if (part[t] == part[s]) H[(size_t)t * ns + s] += u * sg;
        }
    }
    for (int i = 0; i < ns; ++i) for (int j = i + 1; j < ns; ++j) {
        double v = (H[(size_t)i * ns + j] + H[(size_t)j * ns + i]) / 2.0; H[(size_t)i * ns + j] = v; H[(size_t)j * ns + i] = v;
    }
#undef STEP
    free(conn0);
    return H;
}

/* analyze_hamiltonian (Operator.F90:303-442), followed literally: nets of connected states inside every block of the
 * starting partition (303-374), then blocks reached by one operator from the same block are joined, the higher number
 * becoming the lower and the numbers above moving down, restarting after every join (376-434).
 * part[state] is refined in place; returns the number of blocks. */
static int analyze_hamiltonian(int nd, int nsst0, int *part, const double *H, double threshold)
{
    const int nf = 2 * nd, ns = 1 << nf;
    int *s2s = (int *)malloc(sizeof(int) * (size_t)ns), *c2s = (int *)malloc(sizeof(int) * (size_t)ns);
    int *members = (int *)malloc(sizeof(int) * (size_t)ns), *net = (int *)malloc(sizeof(int) * (size_t)ns);
    for (int s = 0; s < ns; ++s) s2s[s] = -1;
    int nsub = -1;
    for (int b = 0; b < nsst0; ++b) {
        int n = 0;
        for (int s = 0; s < ns; ++s) if (part[s] == b) members[n++] = s;      /* block states in ascending order */
        int start = 0;                                                        /* local index the next net starts from */
        while (start >= 0) {
            int len = 1; net[0] = start;
            nsub += 1; s2s[members[start]] = nsub;
            for (int node = 0; node < len; ++node)
                for (int edge = 0; edge < n; ++edge) {
                    double h = H[(size_t)members[net[node]] * ns + members[edge]];
                    if (h != 0.0 && fabs(h) > threshold) {
                        int member = 0;
                        for (int q = 0; q < len; ++q) if (net[q] == edge) { member = 1; break; }
                        if (!member) { net[len++] = edge; s2s[members[edge]] = nsub; }
                    }
                }
            start = -1;
            for (int node = 0; node < n; ++node) if (s2s[members[node]] == -1) { start = node; break; }
        }
    }
    nsub += 1;
    for (int change = 1; change;) {
        change = 0;
        for (int ib = 0; ib < nd && !change; ++ib) for (int is = 0; is < 2 && !change; ++is) for (int ca = 1; ca <= 2 && !change; ++ca) {
            const int bit = ib + nd * is;
            for (int s = 0; s < ns; ++s) {
                c2s[s] = -1;
                if (ca == 1 && !(s >> bit & 1)) c2s[s] = s2s[s | (1 << bit)];
                else if (ca == 2 && (s >> bit & 1)) c2s[s] = s2s[s & ~(1 << bit)];
            }
            for (int i1 = 0; i1 < ns && !change; ++i1) for (int i2 = i1 + 1; i2 < ns; ++i2) {
                if (s2s[i1] == s2s[i2] && c2s[i1] != c2s[i2] && c2s[i1] != -1 && c2s[i2] != -1) {
                    int big = c2s[i1] > c2s[i2] ? c2s[i1] : c2s[i2], small = c2s[i1] > c2s[i2] ? c2s[i2] : c2s[i1];
                    for (int s = 0; s < ns; ++s) {
                        if (s2s[s] == big) s2s[s] = small;
                        if (s2s[s] > big) s2s[s] -= 1;
                    }
                    change = 1;
                    break;
                }
            }
        }
    }
    nsub = 0;
    for (int s = 0; s < ns; ++s) { part[s] = s2s[s]; if (s2s[s] + 1 > nsub) nsub = s2s[s] + 1; }
    free(s2s); free(c2s); free(members); free(net);
    return nsub;
}

ora_locprob *ora_locprob_build(int nd, const double *U, const double *mu_full, const char *qns_in, double epsevec, double epsblock)
{
    const int nf = 2 * nd, ns = 1 << nf;
    char qns[256];
    strncpy(qns, qns_in ? qns_in : "", sizeof(qns) - 1); qns[sizeof(qns) - 1] = 0;
    const int automatic = strstr(qns, "All") != NULL;
    const int use_nt = strstr(qns, "Nt") != NULL, use_sz = strstr(qns, "Szt") != NULL, use_qz = strstr(qns, "Qzt") != NULL;
    const int use_az = strstr(qns, "Azt") != NULL, use_lz = strstr(qns, "Lzt") != NULL, use_jz = strstr(qns, "Jzt") != NULL;

    ora_locprob *lp = (ora_locprob *)calloc(1, sizeof(*lp));
    lp->nd = nd; lp->nflav = nf; lp->nstates = ns;
    lp->sst_of_state = (int *)malloc(sizeof(int) * (size_t)ns);
    /* qns2substates, States.F90:149-184: blocks numbered by first appearance */
    for (int s = 0; s < ns; ++s) lp->sst_of_state[s] = -1;
    int nsst = 0;
    for (int s1 = 0; s1 < ns; ++s1) {
        if (lp->sst_of_state[s1] != -1) continue;
        lp->sst_of_state[s1] = nsst++;
        for (int s2 = s1; s2 < ns; ++s2) {
            if (lp->sst_of_state[s2] != -1) continue;
            if (use_nt && qn_nt(nd, s1) != qn_nt(nd, s2)) continue;
            if (use_sz && !qeq(qn_szt(nd, s1), qn_szt(nd, s2))) continue;
            if (use_qz && !qeq(qn_qzt(nd, s1), qn_qzt(nd, s2))) continue;
            if (use_az && !qeq(qn_azt(nd, s1), qn_azt(nd, s2))) continue;
            if (use_lz && !qeq(qn_lzt(nd, s1), qn_lzt(nd, s2))) continue;
            if (use_jz && !qeq(qn_lzt(nd, s1) + qn_szt(nd, s1), qn_lzt(nd, s2) + qn_szt(nd, s2))) continue;
            lp->sst_of_state[s2] = nsst - 1;
        }
    }
    if (automatic) {                                    /* CTQMC.F90:7971-7990 */
        double big = 0.0;
        for (int i = 0; i < nf * nf; ++i) if (fabs(mu_full[i]) > big) big = fabs(mu_full[i]);
        for (int i = 0; i < nf * nf * nf * nf; ++i) if (fabs(U[i]) > big) big = fabs(U[i]);
        double *Hf = full_hamiltonian(nd, U, mu_full, lp->sst_of_state, nsst);
        nsst = analyze_hamiltonian(nd, nsst, lp->sst_of_state, Hf, epsblock * big);
        free(Hf);
    }
    lp->nsst = nsst;
    /* init_substates, States.F90:98-144 */
    lp->sst_n = (int *)calloc((size_t)nsst, sizeof(int)); lp->sst_off = (int *)calloc((size_t)nsst, sizeof(int));
    lp->states = (int *)malloc(sizeof(int) * (size_t)ns);
    for (int s = 0; s < ns; ++s) lp->sst_n[lp->sst_of_state[s]] += 1;
    for (int b = 1; b < nsst; ++b) lp->sst_off[b] = lp->sst_off[b - 1] + lp->sst_n[b - 1];
    {
        int *fill = (int *)calloc((size_t)nsst, sizeof(int));
        for (int s = 0; s < ns; ++s) { int b = lp->sst_of_state[s]; lp->states[lp->sst_off[b] + fill[b]++] = s; }
        free(fill);
    }
    lp->nsmax = 0;
    for (int b = 0; b < nsst; ++b) if (lp->sst_n[b] > lp->nsmax) lp->nsmax = lp->sst_n[b];
    int *idx_in_block = (int *)malloc(sizeof(int) * (size_t)ns);
    for (int b = 0; b < nsst; ++b) for (int i = 0; i < lp->sst_n[b]; ++i) idx_in_block[lp->states[lp->sst_off[b] + i]] = i;

    /* init_Psi, Operator.F90:115-245: Connect and the occupation-basis matrices (implicit here: one Fock state maps
     * to one Fock state with the sign of the occupied bits below) */
    lp->conn = (int *)malloc(sizeof(int) * (size_t)nsst * nf * 2);
    for (size_t i = 0; i < (size_t)nsst * nf * 2; ++i) lp->conn[i] = -1;
    for (int b = 0; b < nsst; ++b) for (int i = 0; i < lp->sst_n[b]; ++i) {
        int s = lp->states[lp->sst_off[b] + i];
        for (int f = 0; f < nf; ++f) {
            int bit = (f >> 1) + nd * (f & 1);
            int t = s ^ (1 << bit), tb = lp->sst_of_state[t];
            if (tb == b) continue;                                   /* Operator.F90:150 */
            lp->conn[(b * nf + f) * 2 + ((t > s) ? 0 : 1)] = tb;
        }
    }

    /* H_loc per block: oneparticle_hamiltonian (482-542) + u_hamiltonian (1042-1107) + force_hermitian (1110-1125) */
    double **H = (double **)malloc(sizeof(double *) * (size_t)nsst);
    for (int b = 0; b < nsst; ++b) H[b] = (double *)calloc((size_t)lp->sst_n[b] * lp->sst_n[b], sizeof(double));
    for (int f1 = 0; f1 < nf; ++f1) for (int f2 = 0; f2 < nf; ++f2) {
        int b1 = f1 >> 1, s1 = f1 & 1, b2 = f2 >> 1, s2 = f2 & 1;
        double mu = mu_full[((b1 * 2 + s1) * nd + b2) * 2 + s2];
        if (mu == 0.0) continue;
        for (int s = 0; s < ns; ++s) {
            if (!(s >> BIT(f2) & 1)) continue;
            double sg = fermi_sign(s, BIT(f2)); int t = s ^ (1 << BIT(f2));
            if (t >> BIT(f1) & 1) continue;
            sg *= fermi_sign(t, BIT(f1)); t ^= 1 << BIT(f1);
            int b = lp->sst_of_state[s];
            if (lp->sst_of_state[t] != b) continue;
            int n = lp->sst_n[b];
            H[b][idx_in_block[t] * n + idx_in_block[s]] += mu * sg;
        }
    }
    for (int f1 = 0; f1 < nf; ++f1) for (int f2 = 0; f2 < nf; ++f2) for (int f3 = 0; f3 < nf; ++f3) for (int f4 = 0; f4 < nf; ++f4) {
        double u = 0.5 * U[((f1 * nf + f2) * nf + f3) * nf + f4];
        if (u == 0.0) continue;
        for (int s = 0; s < ns; ++s) {            /* c+_1 c+_2 c_4 c_3 |s> */
            int t = s; double sg = 1.0;
            if (!(t >> BIT(f3) & 1)) continue;
            sg *= fermi_sign(t, BIT(f3)); t ^= 1 << BIT(f3);
            if (!(t >> BIT(f4) & 1)) continue;
            sg *= fermi_sign(t, BIT(f4)); t ^= 1 << BIT(f4);
            if (t >> BIT(f2) & 1) continue;
            sg *= fermi_sign(t, BIT(f2)); t ^= 1 << BIT(f2);
            if (t >> BIT(f1) & 1) continue;
            sg *= fermi_sign(t, BIT(f1)); t ^= 1 << BIT(f1);
            int b = lp->sst_of_state[s];
            if (lp->sst_of_state[t] != b) continue;              /* "sst /= sst1: cycle" */
            int n = lp->sst_n[b];
            H[b][idx_in_block[t] * n + idx_in_block[s]] += u * sg;
        }
    }
    for (int b = 0; b < nsst; ++b) {
        int n = lp->sst_n[b];
        for (int i = 0; i < n; ++i) for (int j = i + 1; j < n; ++j) {
            double v = (H[b][i * n + j] + H[b][j * n + i]) / 2.0; H[b][i * n + j] = v; H[b][j * n + i] = v;
        }
    }

    /* diag_Operator (637-670), set_HEVal (754-767), set_HEvec with EPSEVEC and its renormalisation (771-801) */
    lp->eval = (double *)malloc(sizeof(double) * (size_t)ns);
    lp->evec = (double **)malloc(sizeof(double *) * (size_t)nsst);
    for (int b = 0; b < nsst; ++b) {
        int n = lp->sst_n[b];
        lp->evec[b] = (double *)malloc(sizeof(double) * (size_t)n * n);
        sym_eig(H[b], n, lp->eval + lp->sst_off[b], lp->evec[b]);
        for (int i = 0; i < n; ++i) {
            double sum = 0.0;
            for (int j = 0; j < n; ++j) { double *v = &lp->evec[b][j + n * i]; if (fabs(*v) < epsevec) *v = 0.0; sum += *v * *v; }
            for (int j = 0; j < n; ++j) lp->evec[b][j + n * i] /= sum;
        }
        free(H[b]);
    }
    free(H);
    lp->egs = lp->eval[0];
    for (int b = 0; b < nsst; ++b) if (lp->eval[lp->sst_off[b]] < lp->egs) lp->egs = lp->eval[lp->sst_off[b]];

    /* transform_Psis (1130-1168): Op_EB = EVec(target)^T Op EVec(source) */
    lp->op = (double **)calloc((size_t)nsst * nf * 2, sizeof(double *));
    for (int b = 0; b < nsst; ++b) for (int f = 0; f < nf; ++f) for (int ca = 1; ca <= 2; ++ca) {
        int tb = ora_lp_conn(lp, b, f, ca);
        if (tb < 0) continue;
        int n1 = lp->sst_n[b], n2 = lp->sst_n[tb], bit = BIT(f);
        double *P = (double *)calloc((size_t)n2 * n1, sizeof(double));     /* occupation basis, P[i2 + n2*i1] */
        for (int i1 = 0; i1 < n1; ++i1) {
            int s = lp->states[lp->sst_off[b] + i1];
            int occupied = s >> bit & 1;
            if ((ca == 1) == occupied) continue;
            int t = s ^ (1 << bit);
            if (lp->sst_of_state[t] != tb) continue;
            P[idx_in_block[t] + n2 * i1] = fermi_sign(s, bit);
        }
        double *T = (double *)calloc((size_t)n2 * n1, sizeof(double)), *O = (double *)calloc((size_t)n2 * n1, sizeof(double));
        for (int i = 0; i < n2; ++i) for (int j = 0; j < n1; ++j) {        /* T = P * EVec(b) */
            double a = 0.0;
            for (int k = 0; k < n1; ++k) a += P[i + n2 * k] * lp->evec[b][k + n1 * j];
            T[i + n2 * j] = a;
        }
        for (int i = 0; i < n2; ++i) for (int j = 0; j < n1; ++j) {        /* O = EVec(tb)^T * T */
            double a = 0.0;
            for (int k = 0; k < n2; ++k) a += lp->evec[tb][k + n2 * i] * T[k + n2 * j];
            O[i + n2 * j] = a;
        }
        lp->op[(b * nf + f) * 2 + (ca - 1)] = O;
        free(P); free(T);
    }
#undef BIT
    free(idx_in_block);
    return lp;
}

void ora_locprob_free(ora_locprob *lp)
{
    if (!lp) return;
    for (int b = 0; b < lp->nsst; ++b) free(lp->evec[b]);
    for (size_t i = 0; i < (size_t)lp->nsst * lp->nflav * 2; ++i) free(lp->op[i]);
    free(lp->evec); free(lp->op); free(lp->conn); free(lp->eval); free(lp->states); free(lp->sst_n); free(lp->sst_off);
    free(lp->sst_of_state); free(lp);
}
