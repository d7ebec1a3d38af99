// nus_k_lanczos_pq.hip -- the P/Q register-window resize kernel (nus_k_lanczos_pq.hpp): the factors with Q <= 4 and the dispatch.
#include "nus_k_lanczos_pq.hpp"

namespace nus {

bool lanczos_pq_supported(uint32_t P, uint32_t Q)
{
    return (P == 5 && Q == 4) || (P == 6 && Q == 5) || (P == 5 && Q == 3) || (P == 5 && Q == 2) || (P == 7 && Q == 2) ||
           (P == 7 && Q == 5) || (P == 8 && Q == 5) || (P == 9 && Q == 5);
}

uint32_t lanczos_pq_strip_cols(uint32_t P, uint32_t Q)
{
    if (P == 5 && Q == 4) return PqGeom<5, 4>::kStripCols;
    if (P == 6 && Q == 5) return PqGeom<6, 5>::kStripCols;
    if (P == 5 && Q == 3) return PqGeom<5, 3>::kStripCols;
    if (P == 5 && Q == 2) return PqGeom<5, 2>::kStripCols;
    if (P == 7 && Q == 2) return PqGeom<7, 2>::kStripCols;
    if (P == 7 && Q == 5) return PqGeom<7, 5>::kStripCols;
    if (P == 8 && Q == 5) return PqGeom<8, 5>::kStripCols;
    if (P == 9 && Q == 5) return PqGeom<9, 5>::kStripCols;
    return 0;
}

hipError_t launch_lanczos_pq(const UpscaleLaunch &L, const DeviceTables &T, bool exact, uint32_t P, uint32_t Q, uint32_t rows_per_wave,
                             bool narrow)
{
    if (!lanczos_pq_supported(P, Q) || (uint64_t)L.ow * Q != (uint64_t)L.iw * P || (uint64_t)L.oh * Q != (uint64_t)L.ih * P ||
        (L.iw % Q) != 0 || (L.ih % Q) != 0 || (L.ow % 4) != 0 || !T.lz_wx6 || !T.lz_wy6)
        return hipErrorInvalidValue;
    if (P == 5 && Q == 4) return launch_pq<5, 4>(L, T, exact, rows_per_wave, narrow);
    if (P == 6 && Q == 5) return launch_lanczos_pq_65(L, T, exact, rows_per_wave, narrow);
    if (P == 5 && Q == 3) return launch_pq<5, 3>(L, T, exact, rows_per_wave, narrow);
    if (P == 7 && Q == 2) return launch_pq<7, 2>(L, T, exact, rows_per_wave, narrow);
    if (P == 7 && Q == 5) return launch_lanczos_pq_75(L, T, exact, rows_per_wave, narrow);
    if (P == 8 && Q == 5) return launch_lanczos_pq_85(L, T, exact, rows_per_wave, narrow);
    if (P == 9 && Q == 5) return launch_lanczos_pq_95(L, T, exact, rows_per_wave, narrow);
    return launch_pq<5, 2>(L, T, exact, rows_per_wave, narrow);
}

} // namespace nus
