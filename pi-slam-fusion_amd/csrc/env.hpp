// env.hpp -- which environment variables the library reads.
//
// The PRODUCT library (libpifusion.so) reads four, all documented in include/pifusion.h:
//   PF_CULL=0          render every tile of every keyframe's canvas, as the reference does (same mosaic; pf_set_cull does the same per map)
//   PF_ROCTX=1         the reference's named host sections as roctx ranges (rocprofv3 --marker-trace)
//   PF_DIST_VERIFY=1   the seam exchange checks every packed strip against its source tile after the transfer
//   PF_COPY_THREADS=n  host threads that move blend / save results from the pinned staging ring into a pageable caller buffer
// Every other switch -- A/B partners of a kernel or host form, timing-only ablations that produce wrong tiles, statistics -- exists in the
// EXPERIMENTS library only (libpifusion_exp.so, -DPF_EXPERIMENTS=1; tests and tools load it through PF_LIB): exp_env() is a null constant
// in the product build, so the code behind a switch is dead there and its name is not among the library's strings.
#pragma once
#include <cstdlib>

#ifndef PF_EXPERIMENTS
#define PF_EXPERIMENTS 0
#endif

#if PF_EXPERIMENTS
#define exp_env(name) (std::getenv(name))
inline int    exp_env_int_(const char* n, int d)       { const char* v = std::getenv(n); return v ? std::atoi(v) : d; }
inline double exp_env_double_(const char* n, double d) { const char* v = std::getenv(n); return v ? std::atof(v) : d; }
#define exp_env_int(name, dflt) (exp_env_int_(name, dflt))
#define exp_env_double(name, dflt) (exp_env_double_(name, dflt))
#else
#define exp_env(name) (static_cast<const char*>(nullptr))
#define exp_env_int(name, dflt) (dflt)
#define exp_env_double(name, dflt) (dflt)
#endif
