// nus_k_lanczos_pq_75.hip -- k_lanczos3_pq at x7/5 (nus_k_lanczos_pq.hpp), a translation unit of its own for the build's sake.
#include "nus_k_lanczos_pq.hpp"

namespace nus {

hipError_t launch_lanczos_pq_75(const UpscaleLaunch &L, const DeviceTables &T, bool exact, uint32_t rows_per_wave, bool narrow)
{
    return launch_pq<7, 5>(L, T, exact, rows_per_wave, narrow);
}

} // namespace nus
