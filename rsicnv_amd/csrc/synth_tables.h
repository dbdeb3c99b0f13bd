// synth_tables.h -- host-side construction of the sampler tables used by synth_core.h.
#pragma once
#include <stdint.h>
#include <vector>
#include "synth_core.h"
#include "../../include/rsi_synth.h"

struct SynthTables {
  std::vector<uint32_t> wave;      // RSI_SYNTH_WAVE GC-probability thresholds (x 2^32)
  std::vector<uint64_t> thr;       // concatenated inverse-CDF thresholds (x 2^53)
  std::vector<int32_t> off;        // class -> offset into thr; size = classes + 1
  int gc_levels;                   // 1 (model 0) or RSI_SYNTH_GC_LEVELS (model 1)
};

// Builds the tables for a spec.  IEEE + - * / only, fixed summation order: the result is the same
// on any host.
void synth_build_tables(const rsi_synth_spec& spec, SynthTables& T);
