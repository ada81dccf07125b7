// TEST INFRASTRUCTURE - builds oracle/_ref/libref_gcopter_gpu.so: the reference's GPU-flavoured optimiser
// (/root/reference/src/plan_manage/include/se3gcopter/se3gcopter_gpu.hpp, compiled UNMODIFIED where it lies, against oracle/eigen_shim) with its
// `class cuda_computer` supplied by the drop-in oracle/frx_dropin/cuda_computer.cuh, i.e. by libfrx.so: the reference's own MINCO_S3::addTimeIntPenalty
// call site (se3gcopter_gpu.hpp:219-227) then runs the HIP penalty integrator on the MI355X.  Same exports as ref_gcopter_wrap.cpp under refgpu_* names.
// Hidden visibility + -Bsymbolic: SE3GCOPTER / MINCO_S3 of this library have another layout than those of libref_gcopter.so, and both are loaded by one test.
#define REF_FLAVOUR_GPU 1
#include "ref_gcopter_wrap.cpp"
