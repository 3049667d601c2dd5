/*
 * include/compat/LPCNet/include/lpcnet.h -- the xiph header name the reference's Cython binding includes
 * (extensions/lpcnet/cLPCNet.pxd:1 `cdef extern from "LPCNet/include/lpcnet.h"`), served by libdss_hip.so.
 *
 * Put include/compat on the include path and link -ldss_hip: the reference's extensions/lpcnet/LPCNet.pyx then
 * compiles and links UNCHANGED (tests/test_cpu_boundary.py::test_reference_pyx_links_against_the_library).
 * Every prototype below is declared in include/dss_hip.h, which cites the cLPCNet.pxd line it replaces.
 */
#ifndef DSS_COMPAT_LPCNET_H
#define DSS_COMPAT_LPCNET_H
#include "../../../dss_hip.h"
#define NB_FEATURES 20
#define NB_TOTAL_FEATURES 36
#define LPCNET_FRAME_SIZE 160
#endif
