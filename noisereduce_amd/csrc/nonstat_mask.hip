// Instantiations of the register-tile mask kernels (nonstat_mask.hpp) and their launchers: a translation unit of its
// own so that __graft_entry__.build() compiles it beside api.hip (the 68 kernel bodies take as long as the rest).
#include "nonstat_mask.hpp"

namespace sg {

hipError_t launch_iir_mask(int nt, dim3 grid, hipStream_t st, const float* mag, const double* carry, Geom g, NsTiling tl,
                           double b, double nthresh, double slope, int nf, float p, float* M) {
  switch (nt) {
#define SG_NS_CASE(NT_)                                                                                              \
  case NT_:                                                                                                          \
    hipLaunchKernelGGL(k_iir_mask<NT_>, grid, dim3(256), 0, st, mag, carry, g, tl, b, nthresh, slope, nf, p, M);    \
    return hipGetLastError();
    SG_NS_CASE(0) SG_NS_CASE(1) SG_NS_CASE(2) SG_NS_CASE(3) SG_NS_CASE(4) SG_NS_CASE(5) SG_NS_CASE(6) SG_NS_CASE(7)
    SG_NS_CASE(8) SG_NS_CASE(9) SG_NS_CASE(10) SG_NS_CASE(11) SG_NS_CASE(12) SG_NS_CASE(13) SG_NS_CASE(14)
    SG_NS_CASE(15) SG_NS_CASE(16) SG_NS_CASE(17) SG_NS_CASE(18) SG_NS_CASE(19) SG_NS_CASE(20)
    SG_NS_CASE(25) SG_NS_CASE(34) SG_NS_CASE(37)   // ns_iir_nt_ok
#undef SG_NS_CASE
    default: return hipErrorInvalidValue;
  }
}

hipError_t launch_box_mask(int nt, int kbox, dim3 grid, hipStream_t st, const float* mag, Geom g, double nthresh,
                           double slope, int nf, float p, float* M, int64_t k0) {
  if (kbox != NS_BOX_KB) return hipErrorInvalidValue;
  switch (nt) {
#define SG_BOX_CASE(NT_)                                                                                             \
  case NT_:                                                                                                          \
    hipLaunchKernelGGL((k_box_mask<NT_, NS_BOX_KB>), grid, dim3(256), 0, st, mag, g, nthresh, slope, nf, p, M, k0); \
    return hipGetLastError();
    SG_BOX_CASE(0) SG_BOX_CASE(1) SG_BOX_CASE(2) SG_BOX_CASE(3) SG_BOX_CASE(4) SG_BOX_CASE(5) SG_BOX_CASE(6)
    SG_BOX_CASE(7) SG_BOX_CASE(8) SG_BOX_CASE(9) SG_BOX_CASE(10) SG_BOX_CASE(11) SG_BOX_CASE(12)
#undef SG_BOX_CASE
    default: return hipErrorInvalidValue;
  }
}

}  // namespace sg
