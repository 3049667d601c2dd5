/*
 * include/compat/LPCNet/src/lpcnet_private.h -- second header the reference's binding names
 * (extensions/lpcnet/cLPCNet.pxd:22-23).  It declares decode_packet, which nothing in the reference calls
 * (SURVEY.md 8b), so a declaration is all the generated C needs; libdss_hip.so does not define it.
 */
#ifndef DSS_COMPAT_LPCNET_PRIVATE_H
#define DSS_COMPAT_LPCNET_PRIVATE_H
#include "../include/lpcnet.h"
#ifdef __cplusplus
extern "C" {
#endif
void decode_packet(float features[4][NB_TOTAL_FEATURES], float *vq_mem, const unsigned char buf[8]);
#ifdef __cplusplus
}
#endif
#endif
