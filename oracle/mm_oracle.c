/* SPDX-License-Identifier: GPL-3.0-or-later */
/*
 * mm_oracle.c -- CPU restatement of the monkey-moore relative-search hot path.
 *
 * TEST INFRASTRUCTURE ONLY (see mm_oracle.h).  Parity status: PINNED against the
 * reference's own known-answer tests and against the compiled reference
 * (oracle/_ref) -- tests/test_oracle.py.
 *
 * Plain C99, scalar, single threaded.  Citations are file:line under
 * /root/reference.
 */
#include "mm_oracle.h"

#include <ctype.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

enum { MODE_SIMPLE = 1, MODE_WILDCARD = 2, MODE_VALUE_SCAN = 3 };

struct mmo_plan {
   int elem_bytes;
   int mode;
   int L;
   int max_val;              /* numeric_limits<Ty>::max() */
   long skip_len;            /* 2 * (max + 1), monkey_moore.cpp:63 */
   uint32_t wildcard;

   uint32_t *keyword;
   uint32_t *seq;
   int nseq;

   int *expected_diff;
   int *skip;

   /* wildcard path state, monkey_moore.cpp:144-304 */
   uint32_t *norm;           /* case_normalized_keyword */
   unsigned char *is_literal;
   int *bridge;              /* wc_bridge_offset */
   uint32_t *wc_expected;    /* wc_expected_diff, already truncated to Ty */
   uint32_t *wc_mask;        /* wc_bitmask */
   unsigned char *wst;       /* wildcard_skip_table */
   int has_case_change;
   int mostly_lowercase;
};

static void set_err(char *err, int cap, const char *msg)
{
   if (err && cap > 0) {
      snprintf(err, (size_t)cap, "%s", msg);
   }
}

/* include/mmoore/text_utils.hpp:44-53 */
static int ascii_upper(uint32_t c) { return c < 128 && isupper((unsigned char)c); }
static int ascii_lower(uint32_t c) { return c < 128 && islower((unsigned char)c); }

/* monkey_moore.cpp:86-89: std::map<CharType,int>; later duplicates overwrite
 * earlier ones, operator[] on a missing key yields 0 (:239, :387, :578). */
static int seq_index(const mmo_plan *p, uint32_t c)
{
   for (int i = p->nseq - 1; i >= 0; i--) {
      if (p->seq[i] == c) {
         return i;
      }
   }
   return 0;
}

/* monkey_moore.cpp:551-585.  CharType arithmetic is unsigned 32-bit, the result
 * is stored in an int. */
static void relative_values(const mmo_plan *p, const uint32_t *src, int *dst)
{
   int L = p->L;
   if (p->nseq == 0) {
      dst[0] = (int)(uint32_t)(src[0] - src[L - 1]);
      for (int i = L - 1; i > 0; i--) {
         dst[i] = (int)(uint32_t)(src[i] - src[i - 1]);
      }
   }
   else {
      dst[0] = seq_index(p, src[0]) - seq_index(p, src[L - 1]);
      for (int i = L - 1; i > 0; i--) {
         dst[i] = seq_index(p, src[i]) - seq_index(p, src[i - 1]);
      }
   }
}

/* monkey_moore.cpp:106-142 */
static int preprocess_no_wildcards(mmo_plan *p, char *err, int cap)
{
   int L = p->L;
   relative_values(p, p->keyword, p->expected_diff);

   for (long i = 0; i < p->skip_len; i++) {
      p->skip[i] = L - 1;
   }
   for (int i = L - 1; i >= 0; i--) {
      long index = (long)p->expected_diff[i] + p->max_val;
      if (index >= 0 && index < p->skip_len) {
         if (p->skip[index] == L - 1) {
            p->skip[index] = L - i - 1;
         }
      }
      else {
         set_err(err, cap, "Skip table index out of bounds");
         return -1;
      }
   }
   return 0;
}

/* monkey_moore.cpp:144-304 */
static int preprocess_with_wildcards(mmo_plan *p, char *err, int cap)
{
   int L = p->L;
   memcpy(p->norm, p->keyword, sizeof(uint32_t) * (size_t)L);

   /* Step 1, :150-181 */
   if (p->nseq == 0) {
      int upper = 0, lower = 0;
      for (int i = 0; i < L; i++) {
         upper += ascii_upper(p->keyword[i]);
         lower += ascii_lower(p->keyword[i]);
      }
      p->mostly_lowercase = lower > upper;
      if (upper > 0 && lower > 0) {
         for (int i = 0; i < L; i++) {
            if (upper > lower ? ascii_lower(p->norm[i]) : ascii_upper(p->norm[i])) {
               p->norm[i] = p->wildcard;
            }
         }
      }
   }

   /* Step 2, :185-199 */
   int *valid = (int *)malloc(sizeof(int) * (size_t)L);
   int nvalid = 0;
   for (int i = 0; i < L; i++) {
      p->is_literal[i] = (unsigned char)(p->norm[i] != p->wildcard);
      if (p->is_literal[i]) {
         valid[nvalid++] = i;
      }
   }

   /* Step 3, :203-247 */
   for (int i = 0; i < L; i++) {
      p->expected_diff[i] = 0;
      p->bridge[i] = 0;
      p->wc_expected[i] = 0;
      p->wc_mask[i] = 0;
   }
   uint32_t ty_mask = (uint32_t)p->max_val;
   for (int k = 0; k < nvalid; k++) {
      int cur = valid[k];
      int prev = (k == 0) ? valid[nvalid - 1] : valid[k - 1];
      p->bridge[cur] = prev - cur;
      int rel;
      if (p->nseq == 0) {
         rel = (int)(uint32_t)(p->norm[cur] - p->norm[prev]);
      }
      else {
         rel = seq_index(p, p->norm[cur]) - seq_index(p, p->norm[prev]);
      }
      p->expected_diff[cur] = rel;
      p->wc_expected[cur] = (uint32_t)rel & ty_mask;
      p->wc_mask[cur] = ty_mask;
   }
   free(valid);

   /* Step 4, :250-276 */
   for (long i = 0; i < p->skip_len; i++) {
      p->skip[i] = (int)(signed char)(L - 1);
   }
   for (int i = L - 1; i > 0; --i) {
      long index = (long)p->expected_diff[i] + p->max_val;
      if (index >= 0 && index < p->skip_len) {
         int remaining = 0;
         for (int j = i + 1; j < L; j++) {
            remaining += (p->norm[j] == p->wildcard);
         }
         p->skip[index] = (int)(signed char)(L - remaining - i - 1);
      }
      else {
         set_err(err, cap, "Skip table index out of bounds");
         return -1;
      }
   }

   /* Step 5, :280-303 */
   for (int i = L - 1; i >= 0; --i) {
      if (p->norm[i] == p->wildcard) {
         p->wst[i] = 1;
      }
      else {
         int last = -1;
         for (int j = 0; j < i; j++) {
            if (p->norm[j] == p->wildcard) {
               last = j;
            }
         }
         if (last == -1) {
            last = 0;
         }
         int v = i - last - 1;
         p->wst[i] = (unsigned char)(v > 1 ? v : 1);
      }
   }
   return 0;
}

static int leading_wildcards(const mmo_plan *p)
{
   int n = 0;
   while (n < p->L && p->norm[n] == p->wildcard) {
      n++;
   }
   return n;
}

static mmo_plan *plan_alloc(int elem_bytes, int L, int nseq)
{
   mmo_plan *p = (mmo_plan *)calloc(1, sizeof(*p));
   p->elem_bytes = elem_bytes;
   p->L = L;
   p->max_val = elem_bytes == 1 ? 0xFF : 0xFFFF;
   p->skip_len = ((long)p->max_val + 1) * 2;
   p->nseq = nseq;
   p->keyword = (uint32_t *)calloc((size_t)L, sizeof(uint32_t));
   p->seq = (uint32_t *)calloc((size_t)(nseq > 0 ? nseq : 1), sizeof(uint32_t));
   p->expected_diff = (int *)calloc((size_t)L, sizeof(int));
   p->skip = (int *)calloc((size_t)p->skip_len, sizeof(int));
   p->norm = (uint32_t *)calloc((size_t)L, sizeof(uint32_t));
   p->is_literal = (unsigned char *)calloc((size_t)L, 1);
   p->bridge = (int *)calloc((size_t)L, sizeof(int));
   p->wc_expected = (uint32_t *)calloc((size_t)L, sizeof(uint32_t));
   p->wc_mask = (uint32_t *)calloc((size_t)L, sizeof(uint32_t));
   p->wst = (unsigned char *)calloc((size_t)L, 1);
   return p;
}

void mmo_plan_free(mmo_plan *p)
{
   if (!p) {
      return;
   }
   free(p->keyword); free(p->seq); free(p->expected_diff); free(p->skip);
   free(p->norm); free(p->is_literal); free(p->bridge); free(p->wc_expected);
   free(p->wc_mask); free(p->wst);
   free(p);
}

static mmo_plan *finish_plan(mmo_plan *p, char *err, int cap)
{
   int rc = (p->mode == MODE_WILDCARD) ? preprocess_with_wildcards(p, err, cap)
                                       : preprocess_no_wildcards(p, err, cap);
   if (rc != 0) {
      mmo_plan_free(p);
      return NULL;
   }
   /* Jumps of zero or less never terminate in the reference (monkey_moore.cpp:398,
    * :526-527); the oracle refuses them instead of hanging. */
   int match_jump = (p->mode == MODE_WILDCARD) ? p->L - 1 - leading_wildcards(p) : p->L - 1;
   if (match_jump < 1) {
      set_err(err, cap, "keyword would make the reference loop forever (match jump < 1)");
      mmo_plan_free(p);
      return NULL;
   }
   return p;
}

mmo_plan *mmo_plan_relative(int elem_bytes, const uint32_t *keyword, int keyword_len,
                            uint32_t wildcard, const uint32_t *char_seq, int char_seq_len,
                            char *err, int err_cap)
{
   if ((elem_bytes != 1 && elem_bytes != 2) || keyword_len <= 0) {
      set_err(err, err_cap, "bad arguments");   /* assert(!keyword.empty()), :18 */
      return NULL;
   }
   mmo_plan *p = plan_alloc(elem_bytes, keyword_len, char_seq_len);
   p->wildcard = wildcard;
   memcpy(p->keyword, keyword, sizeof(uint32_t) * (size_t)keyword_len);
   if (char_seq_len > 0) {
      memcpy(p->seq, char_seq, sizeof(uint32_t) * (size_t)char_seq_len);
   }

   /* initialize(), :54-78 */
   int has_wildcards = 0;
   for (int i = 0; i < keyword_len; i++) {
      has_wildcards |= (keyword[i] == wildcard);
   }
   if (char_seq_len == 0) {
      int upper = 0, lower = 0;
      for (int i = 0; i < keyword_len; i++) {
         upper += ascii_upper(keyword[i]);
         lower += ascii_lower(keyword[i]);
      }
      p->has_case_change = upper > 0 && lower > 0;
   }
   p->mode = (has_wildcards || p->has_case_change) ? MODE_WILDCARD : MODE_SIMPLE;
   return finish_plan(p, err, err_cap);
}

mmo_plan *mmo_plan_value_scan(int elem_bytes, const int16_t *values, int n,
                              char *err, int err_cap)
{
   if ((elem_bytes != 1 && elem_bytes != 2) || n <= 0) {
      set_err(err, err_cap, "bad arguments");   /* assert, :28 */
      return NULL;
   }
   mmo_plan *p = plan_alloc(elem_bytes, n, 0);
   p->wildcard = 0;
   for (int i = 0; i < n; i++) {
      p->keyword[i] = (uint32_t)(int32_t)values[i];   /* static_cast<CharType>(short), :34 */
   }
   p->mode = MODE_VALUE_SCAN;
   return finish_plan(p, err, err_cap);
}

int mmo_plan_is_wildcard_path(const mmo_plan *p) { return p->mode == MODE_WILDCARD; }
int mmo_plan_keyword_len(const mmo_plan *p) { return p->L; }

static inline int load_elem(const mmo_plan *p, const void *data, uint64_t i)
{
   return p->elem_bytes == 1 ? (int)((const uint8_t *)data)[i]
                             : (int)((const uint16_t *)data)[i];
}

#define EMIT(pos) do { if (n < cap && out) out[n] = (pos); n++; } while (0)

/* monkey_moore.cpp:316-410 */
static int64_t search_simple(const mmo_plan *p, const void *data, uint64_t len,
                             uint64_t *out, uint64_t cap)
{
   const int L = p->L;
   uint64_t n = 0;
   uint64_t head = 0;
   while (head + (uint64_t)L <= len) {          /* :347 */
      int failed = 0;
      int mismatched = 0;
      for (int k = L - 1; k > 0; --k) {          /* :354-362 */
         int diff = load_elem(p, data, head + (uint64_t)k) - load_elem(p, data, head + (uint64_t)k - 1);
         if (diff != p->expected_diff[k]) {
            mismatched = diff;
            failed = 1;
            break;
         }
      }
      if (!failed) {                             /* :367-371 */
         int diff = load_elem(p, data, head) - load_elem(p, data, head + (uint64_t)L - 1);
         if (diff != p->expected_diff[0]) {
            mismatched = diff;
            failed = 1;
         }
      }
      if (!failed) {
         EMIT(head);
         head += (uint64_t)(L - 1);              /* :398 */
      }
      else {
         int jump = p->skip[mismatched + p->max_val];   /* :402-403 */
         if (jump < 1) {
            jump = 1;
         }
         head += (uint64_t)jump;
      }
   }
   return (int64_t)n;
}

/* monkey_moore.cpp:425-546 */
static int64_t search_wildcard(const mmo_plan *p, const void *data, uint64_t len,
                               uint64_t *out, uint64_t cap)
{
   const int L = p->L;
   const int lead = leading_wildcards(p);
   const uint32_t ty_mask = (uint32_t)p->max_val;
   uint64_t n = 0;
   uint64_t head = 0;
   while (head + (uint64_t)L <= len) {           /* :449 */
      int matches = 0;
      int mismatched = 0;
      for (; matches < L; matches++) {           /* :453-470 */
         int i = L - matches - 1;
         int cur = load_elem(p, data, head + (uint64_t)i);
         int prev = load_elem(p, data, head + (uint64_t)(i + p->bridge[i]));
         uint32_t cd = (uint32_t)(cur - prev) & ty_mask;
         if ((cd & p->wc_mask[i]) != p->wc_expected[i]) {
            mismatched = cur - prev;
            break;
         }
      }
      if (matches == L) {
         EMIT(head);
         head += (uint64_t)(L - 1 - lead);       /* :526-527 */
      }
      else {
         int jump = p->skip[mismatched + p->max_val];   /* :531-538 */
         if (jump < 1) {
            jump = 1;
         }
         int wj = p->wst[L - matches - 1];
         if (wj < jump) {
            jump = wj;
         }
         head += (uint64_t)jump;
      }
   }
   return (int64_t)n;
}

int64_t mmo_search(const mmo_plan *p, const void *data, uint64_t data_len,
                   uint64_t *out, uint64_t cap)
{
   /* monkey_moore.cpp:46-48 */
   return p->mode == MODE_WILDCARD ? search_wildcard(p, data, data_len, out, cap)
                                   : search_simple(p, data, data_len, out, cap);
}

static int cmp_u64(const void *a, const void *b)
{
   uint64_t x = *(const uint64_t *)a, y = *(const uint64_t *)b;
   return x < y ? -1 : (x > y ? 1 : 0);
}

int64_t mmo_engine(const mmo_plan *p, const uint8_t *file, uint64_t file_size,
                   uint32_t block_size, int big_endian,
                   uint64_t *out, uint64_t cap)
{
   const uint32_t S = (uint32_t)p->elem_bytes;
   /* search_engine.cpp:227-234 */
   const uint32_t overlap = (uint32_t)(p->L - 1) * S;
   const uint32_t full = block_size + overlap;
   const uint64_t num_blocks = (file_size + block_size - 1) / block_size;

   uint64_t n = 0;
   uint64_t tmp_cap = 1024;
   uint64_t *tmp = (uint64_t *)malloc(sizeof(uint64_t) * tmp_cap);
   uint8_t *work = (uint8_t *)malloc((size_t)full + 8);

   for (uint64_t b = 0; b < num_blocks; b++) {
      uint64_t offset = b * (uint64_t)block_size;         /* :242, widened to 64 bit */
      uint64_t remaining = file_size - offset;
      uint32_t size = (uint32_t)(remaining < full ? remaining : full);   /* :245-247 */

      for (uint32_t pad = 0; pad < S; pad++) {            /* :129-133 */
         uint64_t count = size / S;                        /* :137 */
         if ((uint64_t)pad + count * S > size) {           /* :139-141 */
            count -= 1;
         }
         memcpy(work, file + offset + pad, (size_t)(count * S));
         if (S == 2 && big_endian) {                       /* :143-145 on a little-endian host */
            uint16_t *w = (uint16_t *)work;
            for (uint64_t i = 0; i < count; i++) {
               w[i] = (uint16_t)((w[i] << 8) | (w[i] >> 8));   /* byteswap.hpp:26-29 */
            }
         }
         for (;;) {
            int64_t k = mmo_search(p, work, count, tmp, tmp_cap);   /* :147 */
            if ((uint64_t)k <= tmp_cap) {
               for (int64_t i = 0; i < k; i++) {
                  uint64_t off = offset + tmp[i] * S + pad;         /* :151-154 */
                  if (n < cap && out) {
                     out[n] = off;
                  }
                  n++;
               }
               break;
            }
            tmp_cap = (uint64_t)k;
            tmp = (uint64_t *)realloc(tmp, sizeof(uint64_t) * tmp_cap);
         }
      }
   }
   free(tmp);
   free(work);
   if (out) {
      qsort(out, (size_t)(n < cap ? n : cap), sizeof(uint64_t), cmp_u64);   /* :193-197 */
   }
   return (int64_t)n;
}

/* Helper: insert into a sorted (key,value) list like std::map::operator[]=. */
static int map_put(uint32_t *keys, uint32_t *vals, int n, int cap, uint32_t k, uint32_t v)
{
   int i = 0;
   while (i < n && keys[i] < k) {
      i++;
   }
   if (i < n && keys[i] == k) {
      vals[i] = v;
      return n;
   }
   if (n >= cap) {
      return n;
   }
   memmove(keys + i + 1, keys + i, sizeof(uint32_t) * (size_t)(n - i));
   memmove(vals + i + 1, vals + i, sizeof(uint32_t) * (size_t)(n - i));
   keys[i] = k;
   vals[i] = v;
   return n + 1;
}

int mmo_values_map(const mmo_plan *p, const void *at, uint32_t *keys, uint32_t *vals, int cap)
{
   const uint32_t ty_mask = (uint32_t)p->max_val;
   int n = 0;
   if (p->mode == MODE_VALUE_SCAN) {
      return 0;                                            /* monkey_moore.cpp:377 */
   }
   if (p->mode == MODE_SIMPLE) {                           /* :380-392 */
      if (p->nseq == 0) {
         int distance = load_elem(p, at, 0) - (int)p->keyword[0];
         n = map_put(keys, vals, n, cap, 'A', (uint32_t)('A' + distance) & ty_mask);
         n = map_put(keys, vals, n, cap, 'a', (uint32_t)('a' + distance) & ty_mask);
      }
      else {
         int distance = load_elem(p, at, 0) - seq_index(p, p->keyword[0]);
         for (int i = 0; i < p->nseq; i++) {
            n = map_put(keys, vals, n, cap, p->seq[i],
                        (uint32_t)(seq_index(p, p->seq[i]) + distance) & ty_mask);
         }
      }
      return n;
   }
   /* wildcard path, :444-447 and :476-521 */
   int first = 0;
   while (first < p->L && !p->is_literal[first]) {
      first++;
   }
   if (p->nseq == 0) {
      int distance = load_elem(p, at, (uint64_t)first) - (int)p->norm[first];
      if (!p->has_case_change) {
         n = map_put(keys, vals, n, cap, 'A', (uint32_t)('A' + distance) & ty_mask);
         n = map_put(keys, vals, n, cap, 'a', (uint32_t)('a' + distance) & ty_mask);
      }
      else {
         int opp = -1;
         for (int i = 0; i < p->L; i++) {
            if (p->mostly_lowercase ? ascii_upper(p->keyword[i]) : ascii_lower(p->keyword[i])) {
               opp = i;
               break;
            }
         }
         if (opp < 0) {
            return -1;                                     /* :495-497 throws */
         }
         int odist = load_elem(p, at, (uint64_t)opp) - (int)p->keyword[opp];
         n = map_put(keys, vals, n, cap, 'A',
                     (uint32_t)('A' + (p->mostly_lowercase ? odist : distance)) & ty_mask);
         n = map_put(keys, vals, n, cap, 'a',
                     (uint32_t)('a' + (p->mostly_lowercase ? distance : odist)) & ty_mask);
      }
   }
   else {
      int distance = load_elem(p, at, (uint64_t)first) - seq_index(p, p->keyword[first]);
      for (int i = 0; i < p->nseq; i++) {
         n = map_put(keys, vals, n, cap, p->seq[i],
                     (uint32_t)(seq_index(p, p->seq[i]) + distance) & ty_mask);
      }
   }
   return n;
}

/* splitmix64 (Steele, Lea, Flood 2014), counter form. */
uint64_t mmo_synth_word(uint64_t seed, uint64_t k)
{
   uint64_t z = seed + (k + 1) * 0x9E3779B97F4A7C15ULL;
   z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
   z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
   return z ^ (z >> 31);
}

void mmo_synth_fill(uint8_t *dst, uint64_t first_byte, uint64_t nbytes, uint64_t seed)
{
   uint64_t i = 0;
   while (i < nbytes) {
      uint64_t g = first_byte + i;
      uint64_t w = mmo_synth_word(seed, g >> 3);
      for (unsigned b = (unsigned)(g & 7); b < 8 && i < nbytes; b++, i++) {
         dst[i] = (uint8_t)(w >> (8 * b));
      }
   }
}
