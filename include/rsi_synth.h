/* rsi_synth.h -- deterministic synthetic chromosomes (FASTA bytes + per-base depth) for the
 * configurations of BASELINE.json / SURVEY.md section 8(d).  Input generation only: the
 * reference has no counterpart (its README.md:28-49 recipe downloads a real BAM, which needs
 * network).  Host and device generators are bit-identical for the same spec.
 *
 * Model 0: depth[i] ~ Poisson(mean * cn(i)), independent per base             (config 2)
 * Model 1: depth[i] ~ gamma-Poisson(mean * (0.7 + 0.6*gc201(i)/201) * cn(i), size=nb_size)
 *          where gc201(i) = #{G,C} in [i-100, i+100] clipped to the array      (configs 3-5)
 * cn(i) comes from `events` (code = RSI_CN_*), N runs are written as 'N' and get depth 0,
 * `lower` runs are soft-masked (lower-case, hence neither GC nor N for the reference:
 * loaddata.cpp:483, readref.cpp:95).
 */
#ifndef RSI_SYNTH_H
#define RSI_SYNTH_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct rsi_synth_interval_c {
  int64_t beg;   /* inclusive */
  int64_t end;   /* exclusive */
  int32_t code;  /* events: 0=1x 1=0x 2=0.5x 3=1.5x 4=2x; otherwise unused */
  int32_t pad;
} rsi_synth_interval_c;

typedef struct rsi_synth_spec {
  uint64_t seed;
  int64_t n;
  int32_t model;     /* 0 Poisson, 1 gamma-Poisson with GC-dependent mean */
  int32_t n_events;
  int32_t n_nruns;
  int32_t n_lower;
  double mean;
  double nb_size;
  const rsi_synth_interval_c* events;   /* sorted, disjoint */
  const rsi_synth_interval_c* nruns;    /* sorted, disjoint */
  const rsi_synth_interval_c* lower;    /* sorted, disjoint */
} rsi_synth_spec;

/* Host generator: fasta[n] bytes, depth[n] int32.  Returns 0, or a negative error code. */
int rsi_synth_generate_host(const rsi_synth_spec* spec, uint8_t* fasta, int32_t* depth);

/* Device generator (gfx950): d_fasta / d_depth are device pointers; `stream` is a hipStream_t
 * (NULL = default stream).  Synchronises the stream before returning. */
int rsi_synth_generate_device(const rsi_synth_spec* spec, void* d_fasta, void* d_depth, void* stream);

/* Files for the command-line runs of tests and bench.py: depth as "pos<TAB>depth" lines (1-based positions, one
 * comment line first), the chromosome as a FASTA file of 60-base lines with its .fai index next to it.  0 or < 0. */
int rsi_synth_write_depth_text(const char* path, const int32_t* depth, int64_t n);
int rsi_synth_write_fasta(const char* path, const char* chrom, const uint8_t* fasta, int64_t n);

#ifdef __cplusplus
}
#endif
#endif
