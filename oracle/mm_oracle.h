/* SPDX-License-Identifier: GPL-3.0-or-later */
/*
 * mm_oracle.h -- CPU restatement of the monkey-moore relative-search hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under monkey-moore_amd/ (the product) may
 * include, link or call this.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg use it, and there only as the checker.
 *
 * Parity status: PINNED.  The restatement is checked (tests/test_oracle.py)
 * against (a) every known-answer vector the reference's own Catch2 tests hold
 * for this path (tests/golden/kat_*.json, transcribed inputs/outputs of
 * /root/reference/tests/test_monkey_moore.cpp and test_search_engine.cpp) and
 * (b) differential fixtures produced by the reference itself, compiled from
 * /root/reference by oracle/Makefile into oracle/_ref/ and run by
 * oracle/gen_golden.py (tests/golden/diff_*.json).
 *
 * Every function cites the reference file:line it restates
 * (paths relative to /root/reference).
 */
#ifndef MM_ORACLE_H
#define MM_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct mmo_plan mmo_plan;

/* src/core/monkey_moore.cpp:12-22 (+54-78, 83-100): relative-search ctor.
 * keyword/char_seq are UTF-32 code points.  elem_bytes is 1 or 2 (uint8_t /
 * uint16_t instantiations, monkey_moore.cpp:587-588).  Returns NULL and fills
 * err on the conditions where the reference throws ("Skip table index out of
 * bounds", :139/:274) or would never terminate (jump of 0). */
mmo_plan *mmo_plan_relative(int elem_bytes, const uint32_t *keyword, int keyword_len,
                            uint32_t wildcard, const uint32_t *char_seq, int char_seq_len,
                            char *err, int err_cap);

/* src/core/monkey_moore.cpp:24-39: value-scan ctor. */
mmo_plan *mmo_plan_value_scan(int elem_bytes, const int16_t *values, int n,
                              char *err, int err_cap);

void mmo_plan_free(mmo_plan *p);

/* 1 when the plan runs monkey_moore_wc (monkey_moore.cpp:46-48). */
int mmo_plan_is_wildcard_path(const mmo_plan *p);
int mmo_plan_keyword_len(const mmo_plan *p);

/* src/core/monkey_moore.cpp:41-49 -> :316-410 / :425-546.
 * data: elements of elem_bytes each, host byte order.  Writes up to cap element
 * indices (ascending) to out and returns the total number of matches. */
int64_t mmo_search(const mmo_plan *p, const void *data, uint64_t data_len,
                   uint64_t *out, uint64_t cap);

/* src/core/search_engine.cpp:107-168 + :218-253 + :193-197 with the block
 * offset widened to 64 bits (the shipped `i * block_base_size` is a 32-bit
 * multiply, :241-242).  file: raw file bytes.  Returns total number
 * of matches, byte offsets ascending in out. */
int64_t mmo_engine(const mmo_plan *p, const uint8_t *file, uint64_t file_size,
                   uint32_t block_size, int big_endian,
                   uint64_t *out, uint64_t cap);

/* src/core/monkey_moore.cpp:374-393 and :472-521: the per-match equivalency
 * map.  `at` points at the first element of the match (host byte order).
 * Writes up to cap (key,value) pairs ordered by key; returns the pair count. */
int mmo_values_map(const mmo_plan *p, const void *at,
                   uint32_t *keys, uint32_t *vals, int cap);

/* Synthetic ROM generator shared by tests and bench: byte i of the
 * ROM is byte (i & 7), little endian, of splitmix64 word (i >> 3). */
uint64_t mmo_synth_word(uint64_t seed, uint64_t k);
void mmo_synth_fill(uint8_t *dst, uint64_t first_byte, uint64_t nbytes, uint64_t seed);

#ifdef __cplusplus
}
#endif
#endif
