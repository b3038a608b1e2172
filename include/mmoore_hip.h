/* SPDX-License-Identifier: GPL-3.0-or-later */
/*
 * mmoore_hip.h -- C ABI of the MI355X relative-search engine (libmmoore_hip.so).
 *
 * This is the drop-in boundary: plain C, pointers and sizes only, no C++ or
 * torch types.  The C++17 facade in include/mmoore/ (MonkeyMoore<T>::search,
 * SearchEngine<T>::run) is a thin caller of these entry points; any other host
 * language binds the same symbols.
 *
 * Reference interfaces replaced (paths under the reference repository):
 *   mmh_plan_relative / mmh_plan_value_scan
 *        MonkeyMoore<Ty>::MonkeyMoore(...)           include/mmoore/monkey_moore.hpp:30-42
 *        (initialize/preprocess*)                    src/core/monkey_moore.cpp:54-304
 *   mmh_scan with block_bytes == 0
 *        MonkeyMoore<Ty>::search(data, len)          include/mmoore/monkey_moore.hpp:51
 *                                                    src/core/monkey_moore.cpp:316-410, 425-546
 *   mmh_scan with block_bytes > 0
 *        the per-block worker + merge of
 *        SearchEngine<T>::run                        src/core/search_engine.cpp:107-168, 193-197, 218-253
 *        (block offsets are 64 bit here; the reference multiplies in 32 bit, :241-242)
 *
 * All functions return 0 on success, a negative MMH_E_* code on failure and
 * never throw; mmh_last_error() describes the last failure on the calling
 * thread.  There is NO CPU fallback: without a usable HIP device every device
 * entry point fails with MMH_E_DEVICE.
 *
 * Environment (everything the library and the facade read; nothing else changes their behaviour):
 *   MMOORE_HIP_DEVICE        facade: the first device it uses (default 0)
 *   MMOORE_HIP_DEVICES       facade: at most this many devices for one SearchEngine<T>::run (default: all visible)
 *   MMOORE_HIP_DEVICE_LIST   facade: the devices themselves, comma separated (tests name one GPU twice)
 *   MMOORE_HIP_MULTI         facade: 1 = take the multi-device path whatever the file's size
 *   MMOORE_GATHER_TIMEOUT_S  how long mmh_gather_finish waits for a peer rank before it fails (default 120)
 *   MMOORE_SELFTEST          0 = skip the first-use known-answer test of mmh_create ("route health" below)
 *   MMOORE_WARMUP            0 = skip the device warm-up of the first mmh_create (~0.26 s once per process: code objects,
 *                            streams, copy engines, the radix sort); the first scans then pay for it piecemeal
 *   MMOORE_TRACE             diagnostics on stderr: sync, split, lanes, fused, floods, ingest, selftest -- a comma-separated
 *                            list, or 1 for all of them
 *   MMOORE_MAX_CANDIDATES    (tests) where the per-candidate path hands a scan to the flood paths (default 1048576)
 *   MMOORE_TAIL_SUB, MMOORE_TAIL_QUAD_MAXL   (tests) force the tail kernel's candidates-per-wave variant
 */
#ifndef MMOORE_HIP_H
#define MMOORE_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Longest keyword a plan holds.  The reference has no explicit limit, but its wildcard-path tables
 * store keyword_len - 1 in a char (src/core/monkey_moore.cpp:250-253, :270-272): from 129 symbols
 * on that wraps negative and its skip arithmetic changes meaning, so 128 is where parity ends.
 * Every length takes the streaming filter + per-candidate resolvers (beyond 64 symbols the first resolver follows two
 * phases per lane and leaves what it cannot settle to the forward engine, csrc/mm_forward.h, whose phase maps are sized
 * for 127 phases). */
#define MMH_MAX_KEYWORD 128

enum {
   MMH_OK = 0,
   MMH_E_ARG = -1,              /* bad argument */
   MMH_E_PLAN = -2,             /* keyword the reference rejects or cannot terminate on */
   MMH_E_DEVICE = -3,           /* HIP error / no device */
   MMH_E_CAPACITY = -4,         /* out buffer too small; *out_count holds the need */
   MMH_E_STATE = -5,            /* no ROM attached, etc. */
   MMH_E_ABORTED = -6           /* mmh_rom_load_file_watched: the caller's abort word was raised */
};

enum { MMH_MODE_SIMPLE = 1, MMH_MODE_WILDCARD = 2, MMH_MODE_VALUE_SCAN = 3 };

/* Flattened pattern plan (POD).  One compare per keyword position i, visited
 * from i = L-1 down to 0 exactly like the reference loops:
 *    d = (int)x[h+i] - (int)x[h+i+bridge[i]];   mismatch iff ((d ^ expected[i]) & cmp_mask[i]) != 0
 * cmp_mask is 0xFFFFFFFF on the simple / value-scan path (signed, un-wrapped
 * compare, monkey_moore.cpp:336-345), the element mask on literal positions of
 * the wildcard path (modular compare, :457-470) and 0 on wildcard positions.
 * A mismatch at i jumps min(wst[i], max(skip(d), 1)); a match jumps match_jump. */
typedef struct mmh_plan_desc {
   uint32_t elem_bytes;                 /* 1 or 2 */
   uint32_t mode;                       /* MMH_MODE_* */
   uint32_t L;                          /* keyword length, 2..MMH_MAX_KEYWORD */
   uint32_t match_jump;                 /* L-1 (:398) or L-1-leading wildcards (:526) */
   uint32_t lead_wildcards;
   uint32_t first_literal;              /* first non-wildcard position (:444-447) */
   int32_t  default_skip;               /* skip for diffs not listed below */
   uint32_t n_skip;                     /* number of listed diffs */
   int32_t  expected[MMH_MAX_KEYWORD];  /* expected_diff */
   uint32_t cmp_mask[MMH_MAX_KEYWORD];
   int32_t  skip_diff[MMH_MAX_KEYWORD]; /* sparse bad-character table: diff -> raw skip value */
   int32_t  skip_val[MMH_MAX_KEYWORD];
   int8_t   bridge[MMH_MAX_KEYWORD];    /* wc_bridge_offset; -1 / L-1 on the simple path */
   uint8_t  wst[MMH_MAX_KEYWORD];       /* wildcard_skip_table; 255 = no cap */
} mmh_plan_desc;

typedef struct mmh_ctx mmh_ctx;

const char *mmh_last_error(void);

/* ---- pattern plan (host only, no device needed) ------------------------- */

/* keyword / char_seq: UTF-32 code points.  wildcard 0 = the MonkeyMoore default. */
int mmh_plan_relative(uint32_t elem_bytes, const uint32_t *keyword, uint32_t keyword_len,
                      uint32_t wildcard, const uint32_t *char_seq, uint32_t char_seq_len,
                      mmh_plan_desc *out);
int mmh_plan_value_scan(uint32_t elem_bytes, const int16_t *values, uint32_t n, mmh_plan_desc *out);

/* ---- device context ------------------------------------------------------ */

int mmh_device_count(int *count);
int mmh_create(int device, mmh_ctx **out);
void mmh_destroy(mmh_ctx *ctx);

/* Launch on a caller-owned hipStream_t (e.g. torch's current stream); NULL = the
 * context's own stream. */
int mmh_set_stream(mmh_ctx *ctx, void *hip_stream);

/* ROM: either copied from host memory into a library-owned device buffer, or a
 * borrowed device pointer (16-byte aligned, stays valid until detached). */
int mmh_rom_upload(mmh_ctx *ctx, const void *host, uint64_t nbytes);
int mmh_rom_attach(mmh_ctx *ctx, const void *device_ptr, uint64_t nbytes);
int mmh_rom_download(mmh_ctx *ctx, uint64_t first_byte, void *host, uint64_t nbytes);

/* ROM straight from a file: bytes [file_offset, file_offset + nbytes) of `path` become the
 * context's ROM.  `threads` readers (0 = automatic, at most 16) pread() 4 MiB pieces into
 * pinned staging and queue each piece's host-to-device copy behind its read, so file reads
 * and PCIe transfers overlap.  Replaces the workers' per-block ifstream reads of
 * src/core/search_engine.cpp:120-145.  MMH_E_ARG: cannot open / short read. */
int mmh_rom_load_file(mmh_ctx *ctx, const char *path, uint64_t file_offset, uint64_t nbytes, int threads);
/* The same load, watched from another thread -- the reference's dispatcher polls its abort flag between blocks and
 * reports progress per block while the workers read (src/core/search_engine.cpp:161-187); here the bulk of a
 * file search's time is this ingest, so it is what has to be interruptible and observable:
 *   abort_word  (may be NULL) is read every 0.1 ms by the calling thread, which then only supervises the readers; once
 *               it is non-zero the readers are told to stop and the call returns MMH_E_ABORTED at once (measured:
 *               0.1-0.5 ms after the word went up) -- without waiting for readers that sit inside a blocking call or
 *               for copies already queued; those are waited for before the ROM or the staging buffers are used again
 *               (the next load, scan, or mmh_destroy), and that wait is itself called off by the next load's abort
 *               word.  The ROM's contents are undefined after an abort.  Neither word is touched once the call is back;
 *   bytes_done  (may be NULL) is set to 0 at the start and follows the bytes whose read has finished and whose
 *               host-to-device copy has been queued (monotone; nbytes once a load has succeeded).
 * Both words are accessed with relaxed atomic loads / stores; the caller polls bytes_done from its own thread. */
int mmh_rom_load_file_watched(mmh_ctx *ctx, const char *path, uint64_t file_offset, uint64_t nbytes, int threads,
                              const volatile int32_t *abort_word, volatile uint64_t *bytes_done);
/* wall time, size and reader count of the last mmh_rom_load_file */
int mmh_last_load_stats(mmh_ctx *ctx, double *seconds, uint64_t *bytes, int *threads);
/* Packed copy of bytes_each bytes at each of n ROM byte offsets into host_out (n * bytes_each
 * bytes; positions behind the ROM read as 0): the elements under every match, from which the
 * host builds the equivalency maps of src/core/monkey_moore.cpp:374-393 without keeping its
 * own copy of the file. */
int mmh_rom_gather(mmh_ctx *ctx, const uint64_t *rom_offsets, uint64_t n, uint32_t bytes_each, void *host_out);

/* Synthetic ROM (bench/tests): fill the attached/owned ROM [first_byte, +nbytes)
 * with splitmix64 words indexed by (rom_base_offset + byte) / 8, on the device. */
int mmh_rom_alloc(mmh_ctx *ctx, uint64_t nbytes);
int mmh_rom_synth(mmh_ctx *ctx, uint64_t seed, uint64_t rom_base_offset);
int mmh_rom_poke(mmh_ctx *ctx, uint64_t first_byte, const void *host, uint64_t nbytes);
int mmh_rom_fill(mmh_ctx *ctx, uint64_t first_byte, uint64_t nbytes, int value, int ramp);

/* ---- the scan -------------------------------------------------------------
 * block_bytes == 0 : one chain over the whole ROM viewed as elements of
 *                    plan->elem_bytes in device (little-endian) order; results
 *                    are ELEMENT indices (MonkeyMoore<Ty>::search).
 * block_bytes  > 0 : engine semantics -- the chain restarts at every block of
 *                    block_bytes (+ (L-1)*elem_bytes overlap) and, for 16-bit
 *                    elements, at each of the two byte alignments; results are
 *                    BYTE offsets + base_offset (SearchEngine<T>::run).
 * Results are written ascending to out (host memory).  If more than cap exist
 * the call returns MMH_E_CAPACITY with *out_count = the total. */
int mmh_scan(mmh_ctx *ctx, const mmh_plan_desc *plan, uint64_t block_bytes, int big_endian,
             uint64_t base_offset, uint64_t *out, uint64_t cap, uint64_t *out_count);

/* Scans in flight, for back-to-back scans (many keywords, many ROM partitions):
 * mmh_scan_submit enqueues a scan with the arguments of mmh_scan on one of three internal lanes
 * (own workspace and result block, on two streams of the context) and returns a ticket; mmh_scan_collect waits for
 * that scan and delivers its offsets exactly as mmh_scan would.  At most MMH_MAX_IN_FLIGHT
 * tickets may be outstanding (a further submit fails with MMH_E_STATE); collect them in the
 * order they were submitted; the ROM must not change while one is outstanding.  The host's share
 * of a scan and the kernel behind the streaming filter then overlap the next scan's streaming
 * kernel: two outstanding tickets already do that, the third keeps the device fed when the host
 * is late.  MMH_E_CAPACITY from collect leaves the ticket outstanding: collect again with more room.
 * The lanes use two HIP streams of their own; in a process with more than four streams in all, raise the HIP
 * runtime's GPU_MAX_HW_QUEUES (default 4) before HIP initialises, or streams share hardware queues and the
 * scans in flight no longer overlap. */
#define MMH_MAX_IN_FLIGHT 3
int mmh_scan_submit(mmh_ctx *ctx, const mmh_plan_desc *plan, uint64_t block_bytes, int big_endian,
                    uint64_t base_offset, int *ticket);
int mmh_scan_collect(mmh_ctx *ctx, int ticket, uint64_t *out, uint64_t cap, uint64_t *out_count);

/* ---- multi-GPU: block-aligned partitions + RCCL gather of the offset lists ----------------
 * Replaces the reference's thread dispatcher and merge (src/core/search_engine.cpp:66-188,
 * :193-197).  Every SearchBlock is an independent chain, so the file is dealt out in whole
 * blocks and the only communication is the gather of the ascending per-GPU lists (partitions
 * are in rank order: concatenation is globally ascending).  The collective is issued by this
 * library (librccl, ncclAllGather over xGMI) from the device-side copy of the ordered list every
 * scan leaves in HBM; see csrc/mm_multi.hip. */

/* Rank's partition of a file of total_bytes: the blocks [rank*nb/nranks, (rank+1)*nb/nranks)
 * plus (keyword_len-1)*elem_bytes bytes of pattern-length overlap into the next partition
 * (search_engine.cpp:227-230), clipped to the file.  Host only. */
int mmh_partition(uint64_t total_bytes, uint64_t block_bytes, uint32_t keyword_len, uint32_t elem_bytes, int rank,
                  int nranks, uint64_t *first_byte, uint64_t *nbytes);

/* Communicator of a context.  One process per GPU: rank 0 calls mmh_comm_unique_id, the
 * launcher distributes the MMH_COMM_ID_BYTES bytes (torch.distributed store, MPI, a file ...),
 * every rank calls mmh_comm_init_rank (collective: returns when all ranks have joined).
 * One process, several GPUs: mmh_comm_init_all over contexts on distinct devices (context i
 * becomes rank i).  mmh_destroy releases the communicator. */
#define MMH_COMM_ID_BYTES 128
int mmh_comm_unique_id(void *id128);
int mmh_comm_init_rank(mmh_ctx *ctx, const void *id128, int nranks, int rank);
int mmh_comm_init_all(mmh_ctx *const *ctxs, int n);
int mmh_comm_info(mmh_ctx *ctx, int *rank, int *nranks);     /* nranks 0: no communicator */
void mmh_comm_destroy(mmh_ctx *ctx);

/* Gather of the offset lists of all ranks, split in two so that the collective overlaps the
 * next scan: mmh_gather_start enqueues it on the context's communication stream and returns;
 * mmh_gather_finish waits and delivers the concatenation in rank order.  Every rank calls both,
 * in the same order; at most two gathers may be outstanding; finish them oldest first.
 *   offsets == NULL, n == 0 : the list of the most recent mmh_scan of this context, sent from
 *                             HBM without a host round trip (the usual case);
 *   offsets != NULL         : n ascending offsets in host memory.
 *   want_list               : 0 on ranks that only need the total (finish with out == NULL).
 * MMH_E_CAPACITY from finish leaves the gather outstanding: finish again with more room. */
int mmh_gather_start(mmh_ctx *ctx, const uint64_t *offsets, uint64_t n, int want_list);
int mmh_gather_finish(mmh_ctx *ctx, uint64_t *out, uint64_t cap, uint64_t *out_count);
/* of the last finished gather, in ms: [0] collective + packing on the device (HIP events on the
 * communication stream), [1] host time inside mmh_gather_start + mmh_gather_finish */
int mmh_last_gather_timings(mmh_ctx *ctx, float *ms2);

/* One process driving n GPUs: context i (rank i of an mmh_comm_init_all communicator) scans the
 * ROM partition attached to it with base_offsets[i] -- one host thread per device -- then the
 * lists are gathered (one grouped collective) and delivered ascending through out, exactly as
 * mmh_scan delivers one device's.  Engine semantics only make sense here (block_bytes > 0). */
int mmh_scan_multi(mmh_ctx *const *ctxs, int n, const mmh_plan_desc *plan, uint64_t block_bytes, int big_endian,
                   const uint64_t *base_offsets, uint64_t *out, uint64_t cap, uint64_t *out_count);

/* Self-test hook (tests only): what mmh_gather_finish would deliver had an nranks-rank all-gather left
 * `records` -- nranks records of MMH_GATHER_RECORD_WORDS words as a scan publishes them: [0] slots,
 * [4] bit 0 = the slots have holes (~0), [6] = matches + 1, from word 8 the slots -- on the device.
 * Runs the packing kernel on them; *longest = the most a record would have to hold (a rank's list, or its
 * slot count when the list has holes): > 16384 means the second (padded) phase would follow and nothing is delivered.  The development box has one GPU: this is how the merge of many ranks' lists
 * is tested there. */
#define MMH_GATHER_RECORD_WORDS (8 + 16384)
int mmh_selftest_gather_pack(mmh_ctx *ctx, const uint64_t *records, int nranks, uint64_t *out, uint64_t cap,
                             uint64_t *out_count, uint64_t *longest);

/* Engine selection (tests / cross-checks): 0 = auto -- streaming filter + per-candidate
 * resolvers, falling back to the forward "dense" engine for inputs they do not suit;
 * 1 = force the sequential one-lane-per-domain chain kernel; 2 = force the dense engine.
 * All three produce identical results. */
int mmh_set_engine(mmh_ctx *ctx, int engine);

/* Whether scans record the HIP events behind mmh_last_timings / mmh_timing_history (default: 1).  The event at a scan's START costs
 * its first kernel's dispatch ~4.5 us on this stack (hipExtLaunchKernelGGL with a start event: 23.1 -> 18.6 us for a synchronous scan
 * of a 2 MiB ROM); a caller that never asks for timings -- the include/mmoore facade does not -- switches them off.  With 0 the
 * timing calls report 0 for every scan but the single-launch ones of small ROMs, whose figures come from the kernel's own clock. */
int mmh_set_timing(mmh_ctx *ctx, int on);

/* Device timings of the last mmh_scan, in milliseconds (HIP events on the scan's
 * stream): [0] streaming filter kernel(s), [1] everything behind it (resolvers, ordering,
 * publication of the results), [2] 0, [3] total.
 * A synchronous scan of a ROM of >= 1 GiB runs as a pipeline of block-aligned parts whose kernels overlap (mm_capi.hip:
 * scan_split; MMH_ROUTE_NO_SPLIT switches it off): then [2] = the number of parts, [0] = the parts' streaming kernels
 * SUMMED (more than their share of the wall time), [3] = the pipeline's wall time on the host, [1] = 0. */
int mmh_last_timings(mmh_ctx *ctx, float *ms4);
/* The same for the most recent scans (up to 64 are kept), oldest first: streaming-kernel
 * and total device time of each.  Elapsed times are computed here, not during the scans. */
int mmh_timing_history(mmh_ctx *ctx, float *filter_ms, float *total_ms, int cap, int *count);
/* Counters of the last scan: [0] candidates, [1] matches, [2] resolver tiles walked,
 * [3] path taken (0 filter + resolver, 1 sequential engine, 2 filter + resolver + hard resolver,
 * 3 dense engine, 4 filter + resolver, then the dense engine on the domains whose candidates the
 * resolvers could not settle -- [2] is the number of such domains then, 5 candidate flood: the
 * dense engine on the domains that hold most candidates, filter + resolver on the rest -- [2] is
 * the number of flooded domains). */
int mmh_last_counters(mmh_ctx *ctx, uint64_t *c4);
/* ---- route health ------------------------------------------------------------------------------
 * The fast routes of a scan -- zero-copy upload of small buffers (the kernels read pinned host memory over PCIe), the
 * single-launch kernel with its grid barrier, the bucketed candidate store, results announced through a polled word
 * in pinned memory instead of a HIP event -- are cross-checked three ways:
 *   * the first mmh_create on a device runs a known-answer scan (mmh_selftest_kat) through the fast routes, through
 *     the plain event-synchronised kernels and through the sequential chain kernel; a fast route that returns
 *     anything else is switched off for the process, loudly (stderr), and mmh_create fails when even the plain
 *     kernels disagree with the known answer (MMOORE_SELFTEST=0 skips the test);
 *   * every block a polled scan publishes is validated before it is trusted (flag bits, counts against capacities,
 *     no slot still holding the poison the host left there, offsets strictly ascending and inside the ROM); a
 *     violation reruns the scan through the plain kernels and is remembered (sticky) -- never a wrong list in silence;
 *   * mmh_set_route switches routes off at run time, so that a caller (tests/test_gpu_fuzz.py on a mismatch,
 *     a soak run) can scan the same ROM every way in one process and see which routes disagree.
 * MonkeyMoore<Ty>::search at the reference benchmark's sizes (benchmarks/bench_search.cpp:67-105 -> src/core/
 * monkey_moore.cpp:41-49) takes exactly these routes. */
enum {
   MMH_ROUTE_NO_SINGLE_LAUNCH = 1,   /* ROMs of <= 4 MiB: streaming kernel + tail kernel instead of mm_scan_fused */
   MMH_ROUTE_NO_ZERO_COPY = 2,       /* uploads of <= 512 KiB are copied to HBM like any other */
   MMH_ROUTE_NO_BUCKETS = 4,         /* the 64 candidate lists + mm_scan_tail instead of the bucketed store + mm_scan_tail2 */
   MMH_ROUTE_NO_POLLED = 8,          /* mm_resolve + rank kernels, the scan's end waited for on a HIP event */
   MMH_ROUTE_NO_SPLIT = 16           /* ROMs of >= 1 GiB: ONE streaming launch over the whole ROM instead of the pipeline of parts
                                        (measurements of the kernel itself: bench.py's roofline figure, rocprofv3 profiles) */
};
enum {
   MMH_FB_NONE = 0,
   MMH_FB_HEADER = 1,                /* flag bits / counters of the published header contradict each other or the scan */
   MMH_FB_CAPACITY = 2,              /* more slots or matches announced than the block or the scan's limits allow */
   MMH_FB_STALE_SLOT = 3,            /* a slot still held the host's poison 2 ms after the flag word showed */
   MMH_FB_ORDER = 4,                 /* the list is not strictly ascending, or its length is not the announced one */
   MMH_FB_RANGE = 5,                 /* an offset outside the ROM */
   MMH_FB_SELFTEST = 6               /* (process-wide) the first-use self-test switched a route off */
};
/* mask: MMH_ROUTE_* bits to switch OFF on this context from the next upload / scan on (0: all routes on). */
int mmh_set_route(mmh_ctx *ctx, uint32_t mask);
/* h[0] first violation seen on this context (MMH_FB_*, sticky; 0 = none), [1] scans rerun through the plain kernels
 * because of one, [2] result slots that showed after the flag word and were waited for (not a violation), [3] the
 * most recent violation, [4] routes off in effect (context | process), [5] routes the self-test switched off for the
 * process, [6] self-test state of the context's device (0 not run, 1 passed, 2 passed with routes off, 3 failed),
 * [7] scans validated; [8..15] header words 0..7 of the block the last synchronous scan published. */
int mmh_health(mmh_ctx *ctx, uint64_t *h16);
/* The self-test's known answer (host only): a 4133-byte ROM, the keyword "abcde" (8-bit, no wildcards), blocks of
 * 1024 bytes; expected = what the reference's engine reports (tests/test_oracle.py checks that against the oracle). */
int mmh_selftest_kat(uint8_t *rom, uint64_t rom_cap, uint64_t *rom_bytes, uint64_t *expected, uint64_t expected_cap,
                     uint64_t *expected_count);
/* Runs the self-test again on this context's device (a context of its own); *routes_off = what it would switch off;
 * MMH_E_DEVICE when the plain kernels fail it. */
int mmh_selftest_run(int device, uint32_t *routes_off);
/* What this box's HBM delivers to a kernel that only reads (csrc/mm_probe.hip): `reps` timed passes of two pure-read
 * kernels over the context's ROM (>= 64 MiB, in HBM), HIP events around every pass.  *mean_GBps = the better pattern's mean
 * rate -- the measured ceiling bench.py reports beside the data sheet's 8 TB/s --, *best_GBps its
 * fastest pass, *ms_per_pass (may be NULL) its mean duration. */
int mmh_selftest_read_probe(mmh_ctx *ctx, int reps, double *best_GBps, double *mean_GBps, double *ms_per_pass);
/* Tests only: damage the block the next polled scan of this context publishes, on the host, before it is validated --
 * 1 flag bits, 2 a slot left at the poison, 3 two slots swapped, 4 a slot outside the ROM, 5 the announced count. */
int mmh_debug_inject(mmh_ctx *ctx, uint32_t kind);

/* How the streaming filter keys on a plan (host only, no device needed; tests and tuning):
 * info[0] number of SWAR conditions (0 = none: the dense engine runs), [1] anchor keyword
 * position, [2] kernel shape id, [3] 1 when survivors are verified inside the filter kernel,
 * [4+2k], [5+2k] keyword position and gap (1 adjacent, 2 over one wildcard) of condition k < 4. */
int mmh_filter_shape(const mmh_plan_desc *plan, uint32_t *info12);

#ifdef __cplusplus
}
#endif
#endif
