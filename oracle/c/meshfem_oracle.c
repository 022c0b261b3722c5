/*
 * oracle/c/meshfem_oracle.c -- plain-C restatement of the reference's CPU assembly path, used
 * (a) to cross-check the numpy oracle and (b) as bench.py's `cpu_baseline` ("kind": "port").
 *
 * TEST INFRASTRUCTURE ONLY: nothing in meshfem_amd/ links or loads this file.
 * PINNING: same status as oracle/meshfem_oracle.py (see its header): quadrature/shape pieces are
 * pinned by the reference's unit-test goldens, the end-to-end K is "parity unpinned" because the
 * reference cannot be built here (Eigen/SuiteSparse/TBB absent).
 *
 * The loop STRUCTURE follows the reference so that the timing is a fair port:
 *   per_element_stiffness  <- LinearElasticity.hh:165-232 (M(c,d) built per pair, Mgpj per j,
 *                             Quadrature<K,2(Deg-1)>::integrate per (i,j); upper triangle only)
 *   threaded over elements <- LinearElasticity.hh:1447-1452 (tbb::parallel_for) -> OpenMP here
 *   serial triplet push    <- LinearElasticity.hh:1414-1434,1454-1455 (24-byte triplets,
 *                             size_t i,j; double v, SparseMatrices.hh:45-72)
 *   sum_repeated           <- SparseMatrices.hh:280-374 (serial column counting sort, per-column
 *                             sort + merge in parallel, exact zeros dropped)
 *   csc                    <- SparseMatrices.hh:422-447
 * Differences, all in the port's favour: stack arrays instead of the per-call std::vector
 * (LinearElasticity.hh:195), no Eigen expression temporaries.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

static const int EDGE_START[6] = {0, 1, 2, 0, 2, 1}; /* Simplex.hh:43 */
static const int EDGE_END[6] = {1, 2, 0, 3, 3, 3};   /* Simplex.hh:44 */

typedef struct { uint64_t i, j; double v; } triplet_t; /* SparseMatrices.hh:45-72 */

static int flat3(int dim, int i, int j) { /* Flattening.hh:23-27 */
    if (i == j) return i;
    if (i < j) return (dim * (dim + 1) - j * (j - 1)) / 2 - (i + 1);
    return (dim * (dim + 1) - i * (i - 1)) / 2 - (j + 1);
}

int oracle_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* EmbeddedElement.hh:211-231 (tet) / :170-190 (tri). gl: dim x (dim+1), column k = grad lambda_k,
 * stored gl[a*(dim+1)+k]. returns volume. */
static double embed(int dim, const double *P, double *gl) {
    if (dim == 3) {
        const double *p0 = P, *p1 = P + 3, *p2 = P + 6, *p3 = P + 9;
        double a[3], b[3], n[4][3], d[3];
        int k;
        for (k = 0; k < 3; ++k) { a[k] = p3[k] - p1[k]; b[k] = p2[k] - p1[k]; d[k] = p0[k] - p1[k]; }
        n[0][0] = a[1] * b[2] - a[2] * b[1]; n[0][1] = a[2] * b[0] - a[0] * b[2]; n[0][2] = a[0] * b[1] - a[1] * b[0];
        double vol6 = d[0] * n[0][0] + d[1] * n[0][1] + d[2] * n[0][2];
#define CROSS(o, x, y) do { o[0] = x[1] * y[2] - x[2] * y[1]; o[1] = x[2] * y[0] - x[0] * y[2]; o[2] = x[0] * y[1] - x[1] * y[0]; } while (0)
        double e20[3], e30[3], e10[3];
        for (k = 0; k < 3; ++k) { e20[k] = p2[k] - p0[k]; e30[k] = p3[k] - p0[k]; e10[k] = p1[k] - p0[k]; }
        CROSS(n[1], e20, e30); CROSS(n[2], e30, e10); CROSS(n[3], e10, e20);
        for (k = 0; k < 4; ++k) { gl[0 * 4 + k] = n[k][0] / vol6; gl[1 * 4 + k] = n[k][1] / vol6; gl[2 * 4 + k] = n[k][2] / vol6; }
        return vol6 / 6.0;
    } else {
        const double *p0 = P, *p1 = P + 2, *p2 = P + 4;
        double e[3][2] = {{p2[0] - p1[0], p2[1] - p1[1]}, {p0[0] - p2[0], p0[1] - p2[1]}, {p1[0] - p0[0], p1[1] - p0[1]}};
        double dA = e[1][0] * e[2][1] - e[1][1] * e[2][0];
        int k;
        for (k = 0; k < 3; ++k) { gl[0 * 3 + k] = -e[k][1] / dA; gl[1 * 3 + k] = e[k][0] / dA; }
        return dA / 2.0;
    }
}

/* EmbeddedElement::gradPhi(i) nodal values: out[node][a], node < (deg==1 ? 1 : dim+1).  :288-313 */
static void grad_phi_nodal(int dim, int deg, const double *gl, int i, double out[4][3]) {
    int nv = dim + 1, j, a;
    if (deg == 1) { for (a = 0; a < dim; ++a) out[0][a] = gl[a * nv + i]; return; }
    if (i < nv) {
        for (j = 0; j < nv; ++j) for (a = 0; a < dim; ++a) out[j][a] = -gl[a * nv + i];
        for (a = 0; a < dim; ++a) out[i][a] *= -3;
    } else {
        int e = i - nv;
        for (j = 0; j < nv; ++j) for (a = 0; a < dim; ++a) out[j][a] = 0.0;
        for (a = 0; a < dim; ++a) { out[EDGE_START[e]][a] = 4 * gl[a * nv + EDGE_END[e]]; out[EDGE_END[e]][a] = 4 * gl[a * nv + EDGE_START[e]]; }
    }
}

/* quadrature points of the degree-2(deg-1) rule (GaussQuadrature.hh:115-127, 283-295) */
static int quad_rule(int dim, int deg, double pts[4][4], double *w) {
    int nv = dim + 1, q, k;
    if (deg == 1) { for (k = 0; k < nv; ++k) pts[0][k] = 1.0 / nv; *w = 1.0; return 1; }
    if (dim == 3) {
        const double c0 = 0.58541019662496845446, c1 = 0.13819660112501051518;
        for (q = 0; q < 4; ++q) for (k = 0; k < 4; ++k) pts[q][k] = (q == k) ? c0 : c1;
        *w = 0.25; return 4;
    }
    { const double c0 = 2 / 3.0, c1 = 1 / 6.0;
      for (q = 0; q < 3; ++q) for (k = 0; k < 3; ++k) pts[q][k] = (q == k) ? c0 : c1;
      *w = 1 / 3.0; return 3; }
}

/* Element::perElementStiffness (upper triangle; lower left untouched).  LinearElasticity.hh:165-232
 * D: flatLen x flatLen symmetric. Ke: (n*dim)^2 row-major. */
static void per_element_stiffness(int dim, int deg, const double *gl, double vol, const double *D, double *Ke) {
    const int nv = dim + 1, n = (dim == 3) ? (deg == 1 ? 4 : 10) : (deg == 1 ? 3 : 6);
    const int ks = n * dim, fl = dim * (dim + 1) / 2, nnod = deg == 1 ? 1 : nv;
    double gp[10][4][3], pts[4][4], w;
    int nq = quad_rule(dim, deg, pts, &w);
    int c, d, a, b, i, j, nd, q;
    for (i = 0; i < n; ++i) grad_phi_nodal(dim, deg, gl, i, gp[i]);
    for (c = 0; c < dim; ++c)
        for (d = c; d < dim; ++d) {
            double M[3][3];
            for (a = 0; a < dim; ++a) for (b = 0; b < dim; ++b) M[a][b] = D[flat3(dim, a, c) * fl + flat3(dim, d, b)]; /* C(a,c,d,b) :203-205 */
            for (j = 0; j < n; ++j) {
                int vj = j * dim + d;
                double Mg[4][3];
                for (nd = 0; nd < nnod; ++nd) for (a = 0; a < dim; ++a) { double s = 0; for (b = 0; b < dim; ++b) s += M[a][b] * gp[j][nd][b]; Mg[nd][a] = s; }
                for (i = 0; i < n; ++i) {
                    int vi = i * dim + c;
                    if (c == d && vi > vj) continue;
                    double val = 0;
                    for (q = 0; q < nq; ++q) {   /* integrate(grad_phis[i](p) . Mgpj(p)) :221-223 */
                        double gi[3] = {0, 0, 0}, mj[3] = {0, 0, 0};
                        if (deg == 1) { for (a = 0; a < dim; ++a) { gi[a] = gp[i][0][a]; mj[a] = Mg[0][a]; } }
                        else for (nd = 0; nd < nv; ++nd) for (a = 0; a < dim; ++a) { gi[a] += pts[q][nd] * gp[i][nd][a]; mj[a] += pts[q][nd] * Mg[nd][a]; }
                        double dt = 0; for (a = 0; a < dim; ++a) dt += gi[a] * mj[a];
                        val += dt;
                    }
                    val *= w * vol;
                    if (vi <= vj) Ke[vi * ks + vj] = val; else Ke[vj * ks + vi] = val;
                }
            }
        }
}

/* Ke for every element (threaded like LinearElasticity.hh:1447-1452).
 * D: nD tensors (nD == 1 or nElem). KeAll: nElem x ks x ks, lower triangle zero-filled. */
void oracle_element_stiffness(int dim, int deg, int64_t nElem, const int32_t *elemNodes, int npe, const double *vertPos,
                              const double *D, int64_t nD, double *KeAll, double *volOut) {
    const int ks = npe * dim, fl = dim * (dim + 1) / 2;
    int64_t e;
#pragma omp parallel for schedule(static)
    for (e = 0; e < nElem; ++e) {
        double P[12], gl[12];
        int k, a;
        for (k = 0; k <= dim; ++k) for (a = 0; a < dim; ++a) P[k * dim + a] = vertPos[(int64_t)elemNodes[e * npe + k] * dim + a];
        double vol = embed(dim, P, gl);
        if (volOut) volOut[e] = vol;
        double *Ke = KeAll + e * ks * ks;
        memset(Ke, 0, sizeof(double) * ks * ks);
        per_element_stiffness(dim, deg, gl, vol, D + (nD == 1 ? 0 : e * fl * fl), Ke);
    }
}

/* serial accumToSparseMatrix (LinearElasticity.hh:1414-1434). returns nnz pushed. */
int64_t oracle_push_triplets(int dim, int64_t nElem, const int32_t *elemNodes, int npe, const int32_t *dofForNode,
                             const double *KeAll, triplet_t *out) {
    const int ks = npe * dim;
    int64_t nnz = 0, e;
    int i, j, ci, cj;
    for (e = 0; e < nElem; ++e) {
        const double *Ke = KeAll + e * ks * ks;
        for (i = 0; i < npe; ++i) {
            int64_t di = elemNodes[e * npe + i]; if (dofForNode) di = dofForNode[di];
            for (j = 0; j < npe; ++j) {
                int64_t dj = elemNodes[e * npe + j]; if (dofForNode) dj = dofForNode[dj];
                if (di > dj) continue;
                for (ci = 0; ci < dim; ++ci) for (cj = 0; cj < dim; ++cj) {
                    if (dim * di + ci > dim * dj + cj) continue;
                    int row = dim * i + ci, col = dim * j + cj;
                    double val = (row <= col) ? Ke[row * ks + col] : Ke[col * ks + row];
                    out[nnz].i = (uint64_t)(dim * di + ci); out[nnz].j = (uint64_t)(dim * dj + cj); out[nnz].v = val; ++nnz;
                }
            }
        }
    }
    return nnz;
}

static int cmp_row(const void *a, const void *b) {
    uint64_t x = ((const triplet_t *)a)->i, y = ((const triplet_t *)b)->i;
    return (x > y) - (x < y);
}

/* TripletMatrix::sumRepeated (SparseMatrices.hh:280-374): serial counting sort by column, per
 * column sort by row + merge (parallel), compaction with exact zeros dropped. In place; returns
 * the new nnz. colPtr (n+1) receives the CSC column pointers (setFromTMatrix :1402-1414). */
int64_t oracle_sum_repeated(int64_t n, int64_t nnz, triplet_t *nz, int64_t *colPtr) {
    int64_t *start = (int64_t *)calloc((size_t)n + 1, sizeof(int64_t));
    triplet_t *tmp = (triplet_t *)malloc(sizeof(triplet_t) * (size_t)(nnz > 0 ? nnz : 1));
    int64_t k, c;
    for (k = 0; k < nnz; ++k) start[nz[k].j + 1]++;
    for (c = 0; c < n; ++c) start[c + 1] += start[c];
    {
        int64_t *cur = (int64_t *)malloc(sizeof(int64_t) * (size_t)(n + 1));
        memcpy(cur, start, sizeof(int64_t) * (size_t)(n + 1));
        for (k = 0; k < nnz; ++k) tmp[cur[nz[k].j]++] = nz[k];
        free(cur);
    }
    int64_t *cnt = (int64_t *)calloc((size_t)n + 1, sizeof(int64_t));
#pragma omp parallel for schedule(dynamic, 1024)
    for (c = 0; c < n; ++c) {
        int64_t b = start[c], e = start[c + 1], o = b, t;
        if (e - b > 1) qsort(tmp + b, (size_t)(e - b), sizeof(triplet_t), cmp_row);
        for (t = b; t < e;) {
            triplet_t acc = tmp[t]; int64_t t2 = t + 1;
            while (t2 < e && tmp[t2].i == acc.i) { acc.v += tmp[t2].v; ++t2; }
            if (acc.v != 0.0) tmp[o++] = acc;   /* pruneTol = 0 */
            t = t2;
        }
        cnt[c + 1] = o - b;
    }
    colPtr[0] = 0;
    for (c = 0; c < n; ++c) colPtr[c + 1] = colPtr[c] + cnt[c + 1];
    for (c = 0; c < n; ++c) memmove(nz + colPtr[c], tmp + start[c], sizeof(triplet_t) * (size_t)cnt[c + 1]);
    int64_t out = colPtr[n];
    free(cnt); free(tmp); free(start);
    return out;
}

/* Whole reference-style assembly: returns seconds per phase in times[4] = {Ke, push, sumRepeated+CSC, total}
 * and the CSC arrays (caller allocates Ai/Ax with capacity cap; returns nnz or -1 if cap too small). */
#include <time.h>
static double now_s(void) { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec + 1e-9 * ts.tv_nsec; }

int64_t oracle_assemble_csc(int dim, int deg, int64_t nElem, const int32_t *elemNodes, int npe, const double *vertPos,
                            const int32_t *dofForNode, int64_t nDoF, const double *D, int64_t nD,
                            int64_t *Ap, int64_t *Ai, double *Ax, int64_t cap, double *times) {
    const int ks = npe * dim;
    double t0 = now_s();
    double *KeAll = (double *)malloc(sizeof(double) * (size_t)nElem * ks * ks);
    triplet_t *nz = (triplet_t *)malloc(sizeof(triplet_t) * (size_t)nElem * ks * ks);   /* KeSize^2 * nelem :1441-1443 */
    if (!KeAll || !nz) { free(KeAll); free(nz); return -2; }
    oracle_element_stiffness(dim, deg, nElem, elemNodes, npe, vertPos, D, nD, KeAll, 0);
    double t1 = now_s();
    int64_t nnz = oracle_push_triplets(dim, nElem, elemNodes, npe, dofForNode, KeAll, nz);
    double t2 = now_s();
    nnz = oracle_sum_repeated((int64_t)dim * nDoF, nnz, nz, Ap);
    int64_t k;
    if (nnz <= cap) for (k = 0; k < nnz; ++k) { Ai[k] = (int64_t)nz[k].i; Ax[k] = nz[k].v; }
    double t3 = now_s();
    times[0] = t1 - t0; times[1] = t2 - t1; times[2] = t3 - t2; times[3] = t3 - t0;
    free(KeAll); free(nz);
    return nnz <= cap ? nnz : -1;
}

/* A TUNED host assembly beside the port above (bench.py cpu_baseline.tuned): what a CPU code that is free to restructure the reference's
 * loop would do -- the pattern is known (the CSC of the upper triangle the port produced), every thread computes the Ke of its elements on
 * its stack and adds the entries straight into Ax (binary search of the row in the column, atomic add); no KeAll array, no triplets, no
 * sort. Same Ke routine, same upper-triangle convention as oracle_push_triplets. Entries the port pruned as exact zeros are skipped
 * (counted in *missed). Returns seconds. */
double oracle_assemble_fused(int dim, int deg, int64_t nElem, const int32_t *elemNodes, int npe, const double *vertPos,
                             const int32_t *dofForNode, const double *D, int64_t nD, const int64_t *Ap, const int64_t *Ai, double *Ax,
                             int64_t n, int64_t *missed) {
    const int ks = npe * dim, fl = dim * (dim + 1) / 2;
    double t0 = now_s();
    int64_t e, k, miss = 0;
#pragma omp parallel for schedule(static)
    for (k = 0; k < Ap[n]; ++k) Ax[k] = 0.0;
#pragma omp parallel for schedule(static) reduction(+ : miss)
    for (e = 0; e < nElem; ++e) {
        double P[12], gl[12], Ke[30 * 30];
        int a, i, j, ci, cj;
        for (i = 0; i <= dim; ++i) for (a = 0; a < dim; ++a) P[i * dim + a] = vertPos[(int64_t)elemNodes[e * npe + i] * dim + a];
        double vol = embed(dim, P, gl);
        memset(Ke, 0, sizeof(double) * ks * ks);
        per_element_stiffness(dim, deg, gl, vol, D + (nD == 1 ? 0 : e * fl * fl), Ke);
        for (i = 0; i < npe; ++i) {
            int64_t di = elemNodes[e * npe + i]; if (dofForNode) di = dofForNode[di];
            for (j = 0; j < npe; ++j) {
                int64_t dj = elemNodes[e * npe + j]; if (dofForNode) dj = dofForNode[dj];
                if (di > dj) continue;
                for (cj = 0; cj < dim; ++cj) {
                    const int64_t col = dim * dj + cj;
                    int64_t lo = Ap[col], hi = Ap[col + 1];
                    for (ci = 0; ci < dim; ++ci) {
                        const int64_t row = dim * di + ci;
                        if (row > col) continue;
                        const int r = dim * i + ci, c = dim * j + cj;
                        const double val = (r <= c) ? Ke[r * ks + c] : Ke[c * ks + r];
                        int64_t b = lo, t = hi;                 /* first index with Ai >= row */
                        while (b < t) { int64_t m = (b + t) >> 1; if (Ai[m] < row) b = m + 1; else t = m; }
                        if (b < hi && Ai[b] == row) {
#pragma omp atomic
                            Ax[b] += val;
                        } else
                            ++miss;
                    }
                }
            }
        }
    }
    if (missed) *missed = miss;
    return now_s() - t0;
}

/* Extend-add of the multifrontal Cholesky (oracle/direct_solve.py). The parent's frontal matrix is kept as a panel
 * P ((ns + nb) x ns, row-major: the columns of the ns unknowns eliminated at this node) and the Schur-complement block
 * S (nb x nb, row-major). A child's update matrix U (nbc x nbc, symmetric, row-major) is added at the positions loc[]:
 * loc < ns -> an eliminated unknown, else ns + index into the parent's boundary. The block (eliminated row, boundary column)
 * is the transpose of (boundary row, eliminated column) and is not stored. */
void oracle_extend_add(double *P, double *S, int64_t ns, int64_t nb, const double *U, int64_t nbc, const int64_t *loc, int threads) {
    /* loc is increasing (child boundary and parent boundary are both sorted; the parent's own range lies below its boundary):
     * entries [0, k0) land in the panel's own columns, the rest in S. Only the lower triangle (j <= i) is read and written --
     * the fronts are symmetric and potrf / trsm / syrk use the lower parts. threads == 1 (fronts inside a subtree that a worker
     * thread factors on its own): the calling thread does it all; the fronts above the subtrees pass the BLAS thread count. */
    int64_t k0 = 0;
    while (k0 < nbc && loc[k0] < ns) ++k0;
#pragma omp parallel for schedule(dynamic, 8) num_threads(threads) if (threads > 1 && nbc >= 512)
    for (int64_t i = 0; i < nbc; ++i) {
        const int64_t li = loc[i];
        const double *u = U + i * nbc;
        double *prow = P + li * ns;
        const int64_t jo = i < k0 ? i + 1 : k0;          /* own columns: j < k0 and j <= i */
        for (int64_t j = 0; j < jo; ++j) prow[loc[j]] += u[j];
        if (i >= k0) {
            double *srow = S + (li - ns) * nb - ns;
            for (int64_t j = k0; j <= i; ++j) srow[loc[j]] += u[j];
        }
    }
}

/* zero fill on all threads (first touch spread over the NUMA nodes) */
void oracle_zero(double *a, int64_t n) {
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) a[i] = 0.0;
}
