/* Plain C99 client of the C ABI (no C++, no Python): two tets sharing a face, bottom triangle clamped, unit load on the
 * apex. argv[1] = device ordinal; -1 = host-only context: mesh and BC logic work, the first device call must fail with
 * MFH_ERR_HIP (there is no CPU fallback). */
#include "meshfem_hip.h"
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#define CHECK(call)                                                                        \
    do {                                                                                   \
        mfh_status st_ = (call);                                                           \
        if (st_ != MFH_OK) { printf("error %d: %s\n", (int)st_, mfh_last_error(ctx)); return st_ == MFH_ERR_HIP ? 3 : 2; } \
    } while (0)

int main(int argc, char **argv) {
    const int device = argc > 1 ? atoi(argv[1]) : 0;
    mfh_ctx *ctx = NULL;
    if (mfh_create(device, &ctx) != MFH_OK) { printf("no device\n"); return 3; }
    const double V[5 * 3] = {0, 0, 0, 1, 0, 0, 0, 1, 0, 0.3, 0.3, 1.0, 0.3, 0.3, -1.0};
    const int32_t T[2 * 4] = {0, 1, 2, 3, 0, 2, 1, 4};
    CHECK(mfh_mesh_build(ctx, 3, 1, 2, 5, T, V));
    CHECK(mfh_material_isotropic(ctx, 200.0, 0.3));
    int64_t nElem = 0, nNode = 0;
    CHECK(mfh_mesh_sizes(ctx, &nElem, &nNode, NULL, NULL, NULL, NULL, NULL));
    printf("%lld elements, %lld nodes\n", (long long)nElem, (long long)nNode);
    CHECK(mfh_assemble(ctx, MFH_ASSEMBLE_GATHER));          /* first device call */
    const int64_t fixed[9] = {0, 1, 2, 3, 4, 5, 6, 7, 8};   /* nodes 0,1,2 clamped */
    CHECK(mfh_fix_variables(ctx, 9, fixed, NULL));
    double f[15] = {0}, u[15];
    f[3 * 3 + 2] = 1.0;                                     /* pull the apex (node 3) along +z */
    mfh_solve_info info;
    CHECK(mfh_solve(ctx, 1, f, u, 1e-12, 1000, &info));
    printf("apex displacement (%g, %g, %g) after %d iterations; the unloaded apex moves by %g\n", u[9], u[10], u[11], info.iterations,
           fabs(u[12]) + fabs(u[13]) + fabs(u[14]));
    if (!(u[11] > 0) || fabs(u[14]) > 1e-12 || fabs(u[0]) > 0) return 4;
    double Ku[15];
    CHECK(mfh_apply_K(ctx, u, Ku));
    if (fabs(Ku[11] - 1.0) > 1e-9) return 5;               /* equilibrium at the loaded degree of freedom */
    mfh_destroy(ctx);
    printf("plain C client ok\n");
    return 0;
}
