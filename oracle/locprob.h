/* TEST INFRASTRUCTURE ONLY (oracle): the local problem of the impurity - Fock states, quantum-number blocks
 * ("superstates"), block-diagonal H_loc, its eigenbasis, and the creation/annihilation operators in that basis.
 * CPU restatement of States.F90:58-334 and Operator.F90:115-245, 482-542, 637-801, 1042-1168 as they are called
 * from init_solver (CTQMC.F90:7957-8031).
 */
#ifndef ORA_LOCPROB_H
#define ORA_LOCPROB_H

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ora_locprob {
    int nd, nflav, nstates;   /* nstates = 4^Nd */
    int nsst;                 /* number of blocks (DStates%NSStates) */
    int nsmax;                /* largest block (NStatesMax) */
    int *sst_of_state;        /* [nstates] Fock state -> block                                   (StatesSStates(:,2)) */
    int *sst_n, *sst_off;     /* [nsst] block size and offset in block order                     (SubStates%NStates, %Offset) */
    int *states;              /* [nstates] Fock states in block order                            (SubStates%States) */
    double *eval;             /* [nstates] eigenvalues in block order, ascending inside a block  (SubStates%EVal) */
    double **evec;            /* [nsst] n x n, column major: evec[j + n*i] = <Fock j | eigen i>   (SubStates%EVec) */
    int *conn;                /* [nsst][nflav][2]: target block of c+ (0) / c (1) or -1          (SubStates%Connect) */
    double **op;              /* [nsst*nflav*2] operator in the eigenbasis, n_target x n_source column major, or NULL
                                 (SubStates%Psis_EB(orb,spin,ca)%Op) */
    double egs;               /* lowest eigenvalue (Trace.F90:576-583) */
} ora_locprob;

/* umatrix (2Nd)^4 with flavour index 2*orb+spin, muimp_full (Nd,2,Nd,2), qns = the QuantumNumbers string
 * ("Nt Szt Qzt", "Nt Szt Azt", ...; "All" = automatic partitioning, analyze_hamiltonian, Operator.F90:303-442, with threshold
 * epsblock = EPSBLOCK relative to the largest element of U and muimp), epsevec = EPSEVEC.  Returns NULL on failure. */
ora_locprob *ora_locprob_build(int nd, const double *umatrix, const double *muimp_full, const char *qns, double epsevec, double epsblock);
void ora_locprob_free(ora_locprob *lp);

static inline int ora_lp_conn(const ora_locprob *lp, int sst, int f, int ca) { return lp->conn[(sst * lp->nflav + f) * 2 + (ca - 1)]; }
static inline const double *ora_lp_op(const ora_locprob *lp, int sst, int f, int ca) { return lp->op[(sst * lp->nflav + f) * 2 + (ca - 1)]; }

#ifdef __cplusplus
}
#endif
#endif
