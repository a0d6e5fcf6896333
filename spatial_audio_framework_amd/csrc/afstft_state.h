/* afstft_state.h — device-resident filterbank state shared by the afSTFT C API and the operators. */
#pragma once
#include "saf_hip_common.h"

namespace saf {

/* Ping-pong buffers: a launch reads history from one copy and writes the new history to the
 * other, so workgroups of one launch never race on it. */
struct AfState {
    int nInst = 0, nCHin = 0, nCHout = 0, hop = SAF_HOP;
    DevBuf<float> ana[2];   /* [nInst][nCHin][15][hop]    last 15 input hops */
    DevBuf<float> syn[2];   /* [nInst][nCHout][9][2 hop]  last 9 synthesised frames */
    int anaPar = 0, synPar = 0;
    void create(int nInst, int nCHin, int nCHout, int hop = SAF_HOP);
    void clear();                                   /* afSTFTlib_clearBuffers (afSTFT_internal.c:213-235) */
    void channelChange(int newIn, int newOut);      /* afSTFTlib_channelChange (afSTFT_internal.c:158-211) */
};

}  // namespace saf
