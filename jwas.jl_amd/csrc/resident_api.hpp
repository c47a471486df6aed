// resident_api.hpp -- host-side launchers of the resident-sampler sweep (resident.hpp), compiled in their own translation
// unit (resident_launch.hip) so that the two halves of the library build in parallel.
#pragma once
#include "sweep.hpp"
#include "resident.hpp"

namespace jw {

// does a resident sampler exist for this (method, traits)?
bool resident_supported(int method, int ntraits);
// dynamic LDS bytes of the sampler kernel for this configuration (the StepSmem carve of k_block_step)
size_t resident_sampler_lds(int method, int ntraits, int bsz, bool dense, bool lpr_mat);
// one launch per sweep (grid 1) on `stream`
hipError_t launch_sampler_resident(int device, int method, int ntraits, bool dense, const ResidentArgs& R, hipStream_t stream);
// one launch per block on `stream`; exactly one of dcols / pcols is non-NULL.  lds_bytes >= the reduction scratch: a larger
// value keeps one workgroup per CU (the geometry the dense stream was tuned for)
hipError_t launch_update_step(int device, int ntraits, bool coop, const UpdateArgs& U, const DenseCols* dcols, const PackedCols* pcols,
                              const ResidentLink& L, const ResidentHelp& H, unsigned grid, hipStream_t stream);

}  // namespace jw
