// Internal interface between sp3_gemm's dispatcher (gemm.hip) and the lean small-M instances (gemm_sm.hip).
#pragma once
#include "common.h"

// tile id (>= 30) of the lean instance that serves this (validated) descriptor, or -1
int sp3_gemm_sm_tile(const sp3_gemm_desc& d);
// true if that instance also serves PAIRED launches (sp3_gemm2): the q/k/v projections' ROPE instances only
bool sp3_gemm_sm_pairs(const sp3_gemm_desc& d);
// launches d (and, in the same launch, `pair` if non-null: sp3_gemm2) on its lean instance; 0 on success
int sp3_gemm_sm_launch(const sp3_gemm_desc& d, const sp3_gemm_desc* pair, hipStream_t stream);
