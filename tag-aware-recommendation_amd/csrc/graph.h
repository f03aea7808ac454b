// The CSR handle behind `tagrec_graph*` and the device-side views of it, shared by the kernels that walk
// adjacency rows (spmm.hip, routing.hip).
#pragma once

#include "common.h"

// Borrowed CSR arrays + the long-row work list built once by tagrec_graph_create (spmm.hip).
struct tagrec_graph {
  int64_t n_rows, n_cols, nnz;
  const int64_t* rowptr;
  const int32_t* col;
  const float* val;
  int64_t n_long, n_chunks;
  int32_t* long_rows;
  int32_t* long_base;
  int2* chunk_desc;
  float* slab;                // n_chunks x kSlabWidth partial sums, allocated with the handle (scratch of the launch in
  size_t slab_floats;         // flight: one stream per handle)
  bool owns_long;             // false: long_rows / long_base / chunk_desc belong to the graph this one was created like,
                              // or live in a caller-provided workspace
  bool owns_slab;             // false: the slab lives in a caller-provided workspace
  bool deferred = false;      // n_long / n_chunks are upper bounds, unused slots of the work list hold -1 (no host read at creation)
};

namespace tagrec {

constexpr int kWavesPerBlock = 4;
constexpr int kLongRow = 1024;   // rows with more stored entries are cut into chunks
constexpr int kChunk = 512;      // entries per chunk (one wavefront each)
constexpr int kSlabWidth = 256;  // floats per chunk in the partial-sum slab = widest vector kernel

struct GraphView {
  int64_t n_rows;
  const int64_t* rowptr;
  const int32_t* col;
  const float* val;
  int64_t n_cols;           // rows of the gathered operand: what the row flags of that operand are counted against
};

// The FIRST `chunk_blocks` blocks of a row-walking launch take the long-row chunks.
struct LongView {
  const int32_t* long_rows;
  const int2* chunk_desc;   // (index into long_rows, chunk number)
  int64_t n_chunks;
  float* slab;              // [n_chunks, width] partial results
  unsigned chunk_blocks;
};

// Check that g->slab (allocated at creation) holds n_chunks x width floats.
int ensure_slab(const tagrec_graph* g, int width);

// *count = number of non-zero bytes of flags[0..n) (rowops.hip)
int count_flags(const uint8_t* flags, int64_t n, unsigned* count, hipStream_t s);

}  // namespace tagrec
