// libcasmvs_io.so: PNG decoding for the input pipeline (include/casmvs_io.h).
//
// What it replaces: PIL's `Image.open(f).convert("RGB")` per view in the reference's dataset classes
// (datasets/dtu.py:168-170, :113-118).  Written against the formats' specifications - RFC 1950 (zlib container),
// RFC 1951 (deflate), ISO/IEC 15948 (PNG chunks, filters) - not against zlib / libpng sources.
//
// Why its own inflate: the image files of a DTU scan are noisy photographs whose deflate streams are literal-heavy; the
// decode loop below keeps a 64-bit bit buffer that is refilled with one unaligned load, resolves a literal / length
// symbol with one lookup in an 11-bit table (entries carry the base value and the number of extra bits), decodes up to
// three literals per refill and copies matches eight bytes at a time.  The output buffer has slack past its expected end so
// that the hot loop needs one bounds check per iteration.  Measured beside PIL in tools/cpu_png_decode_bench.py.
#include "casmvs_io.h"

#include <atomic>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>
#include <thread>
#include <vector>

#include <emmintrin.h>
#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>

namespace {

thread_local char g_error[256] = "";

int fail(int code, const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_error, sizeof(g_error), fmt, ap);
  va_end(ap);
  return code;
}

inline uint64_t load64(const uint8_t *p) {
  uint64_t v;
  memcpy(&v, p, 8);
  return v;   // x86-64 / little-endian hosts only (checked in casmvs_zlib_inflate)
}
inline void store64(uint8_t *p, uint64_t v) { memcpy(p, &v, 8); }
inline uint32_t be32(const uint8_t *p) { return (uint32_t)p[0] << 24 | (uint32_t)p[1] << 16 | (uint32_t)p[2] << 8 | p[3]; }

// ---- Huffman decode tables ------------------------------------------------------------------------------------------
// entry: bits 0-7 bits to consume at this step, 8-11 number of extra bits (or, for F_SUB, the sub-table's index width),
// 12-15 flags, 16-31 value (literal byte / base length / base distance / sub-table offset)
constexpr uint32_t F_BAD = 0x1000, F_EOB = 0x2000, F_SUB = 0x4000, F_LIT = 0x8000;
constexpr int LIT_BITS = 11, DIST_BITS = 8, PRE_BITS = 7;
constexpr int LIT_CAP = (1 << LIT_BITS) + 288 * 16, DIST_CAP = (1 << DIST_BITS) + 32 * 128, OUT_SLACK = 320, IN_PAD = 64;

constexpr uint16_t kLenBase[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
constexpr uint8_t kLenExtra[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
constexpr uint16_t kDistBase[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
constexpr uint8_t kDistExtra[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
constexpr uint8_t kPrecodeOrder[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};

inline uint32_t litlen_entry(int s) {
  if (s < 256) return F_LIT | (uint32_t)s << 16;
  if (s == 256) return F_EOB;
  if (s < 286) return (uint32_t)kLenBase[s - 257] << 16 | (uint32_t)kLenExtra[s - 257] << 8;
  return F_BAD;
}
inline uint32_t dist_entry(int s) { return s < 30 ? ((uint32_t)kDistBase[s] << 16 | (uint32_t)kDistExtra[s] << 8) : F_BAD; }
inline uint32_t plain_entry(int s) { return (uint32_t)s << 16; }

inline unsigned reverse_bits(unsigned code, int len) {
  unsigned r = 0;
  for (int i = 0; i < len; ++i) r |= ((code >> i) & 1u) << (len - 1 - i);
  return r;
}

// Canonical Huffman code of `n` symbols with lengths lens[] (0 = unused, <= 15) -> two-level LSB-first lookup table.
// Over-subscribed length sets are rejected; incomplete ones leave F_BAD entries that fail when a stream reaches them.
template <class EntryFn>
bool build_table(const uint8_t *lens, int n, int primary, uint32_t *table, int cap, EntryFn entry_of) {
  int count[16] = {0};
  for (int s = 0; s < n; ++s) ++count[lens[s]];
  count[0] = 0;
  int left = 1;
  for (int len = 1; len <= 15; ++len) {
    left = (left << 1) - count[len];
    if (left < 0) return false;
  }
  unsigned next[16], code = 0;
  for (int len = 1; len <= 15; ++len) {
    code = (code + count[len - 1]) << 1;
    next[len] = code;
  }
  const int psize = 1 << primary;
  for (int i = 0; i < psize; ++i) table[i] = F_BAD;
  uint8_t sub_bits[1 << LIT_BITS];
  memset(sub_bits, 0, (size_t)psize);
  uint16_t rcode[288];
  for (int s = 0; s < n; ++s) {
    const int len = lens[s];
    if (!len) continue;
    const unsigned r = reverse_bits(next[len]++, len);
    rcode[s] = (uint16_t)r;
    if (len > primary) {
      const unsigned pre = r & (psize - 1);
      if (sub_bits[pre] < len - primary) sub_bits[pre] = (uint8_t)(len - primary);
    } else {
      const uint32_t e = entry_of(s) | (uint32_t)len;
      for (unsigned k = r; k < (unsigned)psize; k += 1u << len) table[k] = e;
    }
  }
  int next_free = psize;
  for (int s = 0; s < n; ++s) {
    const int len = lens[s];
    if (len <= primary) continue;
    const unsigned r = rcode[s], pre = r & (psize - 1);
    if (!(table[pre] & F_SUB)) {
      const int sb = sub_bits[pre];
      if (next_free + (1 << sb) > cap) return false;
      table[pre] = F_SUB | (uint32_t)sb << 8 | (uint32_t)primary | (uint32_t)next_free << 16;
      for (int i = 0; i < (1 << sb); ++i) table[next_free + i] = F_BAD;
      next_free += 1 << sb;
    }
    const unsigned off = table[pre] >> 16, sb = (table[pre] >> 8) & 15u;
    const int l2 = len - primary;
    const uint32_t e = entry_of(s) | (uint32_t)l2;
    for (unsigned k = r >> primary; k < (1u << sb); k += 1u << l2) table[off + k] = e;
  }
  return true;
}

struct Tables {
  uint32_t lit[LIT_CAP];
  uint32_t dist[DIST_CAP];
};

const Tables &fixed_tables() {   // RFC 1951 3.2.6
  static const Tables *t = [] {
    Tables *p = new Tables;
    uint8_t lens[288];
    for (int i = 0; i < 288; ++i) lens[i] = i < 144 ? 8 : i < 256 ? 9 : i < 280 ? 7 : 8;
    build_table(lens, 288, LIT_BITS, p->lit, LIT_CAP, litlen_entry);
    uint8_t dl[32];
    memset(dl, 5, 32);
    build_table(dl, 32, DIST_BITS, p->dist, DIST_CAP, dist_entry);
    return p;
  }();
  return *t;
}

// ---- inflate --------------------------------------------------------------------------------------------------------
struct Inflater {
  const uint8_t *in, *in_end;   // in_end: end of the real data; IN_PAD readable zero bytes follow it
  uint8_t *out, *out_begin, *out_limit;   // OUT_SLACK writable bytes follow out_limit
  uint64_t bitbuf = 0;
  int bitcnt = 0;
  Tables dyn;

#define CASMVS_REFILL()                         \
  do {                                          \
    bitbuf |= load64(in) << bitcnt;             \
    in += (63 - bitcnt) >> 3;                   \
    bitcnt |= 56;                               \
  } while (0)
#define CASMVS_CONSUME(nbits)    \
  do {                           \
    bitbuf >>= (nbits);          \
    bitcnt -= (int)(nbits);      \
  } while (0)

  int error(const char *what) { return fail(CASMVS_IO_CORRUPT, "inflate: %s (input byte %ld, output byte %ld)", what, (long)(in - in_begin), (long)(out - out_begin)); }
  const uint8_t *in_begin = nullptr;

  int read_dynamic_header() {   // RFC 1951 3.2.7
    CASMVS_REFILL();
    const int hlit = (int)(bitbuf & 31) + 257, hdist = (int)((bitbuf >> 5) & 31) + 1, hclen = (int)((bitbuf >> 10) & 15) + 4;
    CASMVS_CONSUME(14);
    if (hlit > 286 || hdist > 30) return error("too many length or distance symbols");
    uint8_t pl[19] = {0};
    for (int i = 0; i < hclen; ++i) {
      if (bitcnt < 3) CASMVS_REFILL();
      pl[kPrecodeOrder[i]] = (uint8_t)(bitbuf & 7);
      CASMVS_CONSUME(3);
    }
    uint32_t pre[1 << PRE_BITS];
    if (!build_table(pl, 19, PRE_BITS, pre, 1 << PRE_BITS, plain_entry)) return error("invalid code lengths set");
    uint8_t lens[288 + 32];
    int i = 0;
    const int total = hlit + hdist;
    while (i < total) {
      if (in > in_end + 8) return error("unexpected end of stream");
      CASMVS_REFILL();
      const uint32_t e = pre[bitbuf & ((1 << PRE_BITS) - 1)];
      if (e & F_BAD) return error("invalid code lengths set");
      CASMVS_CONSUME(e & 0xff);
      const int sym = (int)(e >> 16);
      if (sym < 16) {
        lens[i++] = (uint8_t)sym;
        continue;
      }
      int rep, val = 0;
      if (sym == 16) {
        if (i == 0) return error("invalid bit length repeat");
        val = lens[i - 1];
        rep = 3 + (int)(bitbuf & 3);
        CASMVS_CONSUME(2);
      } else if (sym == 17) {
        rep = 3 + (int)(bitbuf & 7);
        CASMVS_CONSUME(3);
      } else {
        rep = 11 + (int)(bitbuf & 127);
        CASMVS_CONSUME(7);
      }
      if (i + rep > total) return error("invalid bit length repeat");
      memset(lens + i, val, (size_t)rep);
      i += rep;
    }
    if (lens[256] == 0) return error("invalid code -- missing end-of-block");
    uint8_t ll[288] = {0}, dl[32] = {0};
    memcpy(ll, lens, (size_t)hlit);
    memcpy(dl, lens + hlit, (size_t)hdist);
    if (!build_table(ll, 288, LIT_BITS, dyn.lit, LIT_CAP, litlen_entry)) return error("invalid literal/lengths set");
    if (!build_table(dl, 32, DIST_BITS, dyn.dist, DIST_CAP, dist_entry)) return error("invalid distances set");
    return 0;
  }

  int decode_block(const Tables &t) {
    const uint32_t *lt = t.lit, *dt = t.dist;
    constexpr uint64_t LMASK = (1u << LIT_BITS) - 1, DMASK = (1u << DIST_BITS) - 1;
    for (;;) {
      if (out > out_limit) return error("more data than the image holds");
      if (in > in_end + 8) return error("unexpected end of stream");
      CASMVS_REFILL();   // >= 56 bits
      uint32_t e = lt[bitbuf & LMASK];
      if (__builtin_expect(e & F_LIT, 1)) {   // up to three literals from the primary table (<= 33 bits) before the general symbol (<= 20)
        CASMVS_CONSUME(e & 0xff);
        *out++ = (uint8_t)(e >> 16);
        e = lt[bitbuf & LMASK];
        if (__builtin_expect(e & F_LIT, 1)) {
          CASMVS_CONSUME(e & 0xff);
          *out++ = (uint8_t)(e >> 16);
          e = lt[bitbuf & LMASK];
          if (__builtin_expect(e & F_LIT, 1)) {
            CASMVS_CONSUME(e & 0xff);
            *out++ = (uint8_t)(e >> 16);
            e = lt[bitbuf & LMASK];
          }
        }
      }
      if (e & F_SUB) {
        CASMVS_CONSUME(e & 0xff);
        e = lt[(e >> 16) + (bitbuf & ((1u << ((e >> 8) & 15u)) - 1))];
      }
      if (e & F_LIT) {
        CASMVS_CONSUME(e & 0xff);
        *out++ = (uint8_t)(e >> 16);
        continue;
      }
      if (e & (F_EOB | F_BAD)) {
        if (e & F_BAD) return error("invalid literal/length code");
        CASMVS_CONSUME(e & 0xff);
        return 0;
      }
      CASMVS_CONSUME(e & 0xff);
      const unsigned lx = (e >> 8) & 15u;
      const unsigned length = (e >> 16) + (unsigned)(bitbuf & ((1u << lx) - 1));
      CASMVS_CONSUME(lx);
      CASMVS_REFILL();
      uint32_t d = dt[bitbuf & DMASK];
      if (d & F_SUB) {
        CASMVS_CONSUME(d & 0xff);
        d = dt[(d >> 16) + (bitbuf & ((1u << ((d >> 8) & 15u)) - 1))];
      }
      if (d & F_BAD) return error("invalid distance code");
      CASMVS_CONSUME(d & 0xff);
      const unsigned dx = (d >> 8) & 15u;
      const size_t dist = (d >> 16) + (size_t)(bitbuf & ((1u << dx) - 1));
      CASMVS_CONSUME(dx);
      if (dist > (size_t)(out - out_begin)) return error("invalid distance too far back");
      if (out + length > out_limit) return error("more data than the image holds");
      uint8_t *end = out + length;
      const uint8_t *src = out - dist;
      if (dist >= 8) {
        do {
          store64(out, load64(src));
          out += 8;
          src += 8;
        } while (out < end);
      } else if (dist == 1) {
        const uint64_t v = 0x0101010101010101ull * src[0];
        do {
          store64(out, v);
          out += 8;
        } while (out < end);
      } else {
        do *out++ = *src++;
        while (out < end);
      }
      out = end;
    }
  }

  // Whole deflate stream.  Returns 0 and leaves `in` at the first byte after the stream.
  int run() {
    in_begin = in;
    for (;;) {
      if (in > in_end + 8) return error("unexpected end of stream");
      CASMVS_REFILL();
      const int final_block = (int)(bitbuf & 1), type = (int)((bitbuf >> 1) & 3);
      CASMVS_CONSUME(3);
      if (type == 0) {   // stored: drop the rest of the byte, give back the whole bytes in the buffer
        CASMVS_CONSUME(bitcnt & 7);
        in -= bitcnt >> 3;
        bitbuf = 0;
        bitcnt = 0;
        if (in + 4 > in_end) return error("unexpected end of stream");
        const unsigned len = in[0] | (unsigned)in[1] << 8, nlen = in[2] | (unsigned)in[3] << 8;
        if ((len ^ nlen) != 0xffffu) return error("invalid stored block lengths");
        in += 4;
        if (in + len > in_end) return error("unexpected end of stream");
        if (out + len > out_limit) return error("more data than the image holds");
        memcpy(out, in, len);
        in += len;
        out += len;
      } else if (type == 1) {
        if (int rc = decode_block(fixed_tables())) return rc;
      } else if (type == 2) {
        if (int rc = read_dynamic_header()) return rc;
        if (int rc = decode_block(dyn)) return rc;
      } else {
        return error("invalid block type");
      }
      if (final_block) break;
    }
    in -= bitcnt >> 3;   // unread whole bytes
    bitcnt = 0;
    bitbuf = 0;
    if (in > in_end) return error("unexpected end of stream");
    return 0;
  }
};

// Adler-32 (RFC 1950 8.2) sixteen bytes at a time with SSE2 (part of every x86-64): s1 by sums of absolute differences,
// s2 = sum over blocks of 16 * (s1 before the block) + the bytes weighted 16 .. 1.  5552 bytes = 347 blocks keep every
// 32-bit lane below 2^32 (lane of s1 sums <= 65521 k + 1020 k^2, times 16 = 2.33e9 at k = 347); lanes are added in 64 bits.
uint32_t adler32(const uint8_t *p, size_t n) {
  uint64_t a = 1, b = 0;
  const __m128i zero = _mm_setzero_si128();
  const __m128i w_lo = _mm_set_epi16(9, 10, 11, 12, 13, 14, 15, 16), w_hi = _mm_set_epi16(1, 2, 3, 4, 5, 6, 7, 8);
  while (n >= 16) {
    size_t blocks = n / 16 < 347 ? n / 16 : 347;
    n -= blocks * 16;
    __m128i v_s1 = _mm_cvtsi32_si128((int)a), v_ps = zero, v_s2 = zero;
    for (; blocks; --blocks, p += 16) {
      const __m128i bytes = _mm_loadu_si128((const __m128i *)p);
      v_ps = _mm_add_epi32(v_ps, v_s1);
      v_s1 = _mm_add_epi32(v_s1, _mm_sad_epu8(bytes, zero));
      v_s2 = _mm_add_epi32(v_s2, _mm_madd_epi16(_mm_unpacklo_epi8(bytes, zero), w_lo));
      v_s2 = _mm_add_epi32(v_s2, _mm_madd_epi16(_mm_unpackhi_epi8(bytes, zero), w_hi));
    }
    uint32_t l1[4], lp[4], l2[4];
    _mm_storeu_si128((__m128i *)l1, v_s1);
    _mm_storeu_si128((__m128i *)lp, v_ps);
    _mm_storeu_si128((__m128i *)l2, v_s2);
    const uint64_t s1 = (uint64_t)l1[0] + l1[1] + l1[2] + l1[3], ps = (uint64_t)lp[0] + lp[1] + lp[2] + lp[3];
    const uint64_t s2 = (uint64_t)l2[0] + l2[1] + l2[2] + l2[3];
    b = (b + 16 * ps + s2) % 65521u;
    a = s1 % 65521u;
  }
  while (n--) {
    a += *p++;
    b += a;
  }
  return (uint32_t)(b % 65521u) << 16 | (uint32_t)(a % 65521u);
}

// zlib container around the deflate stream.  `src` must be followed by IN_PAD readable bytes, `dst` by OUT_SLACK writable ones.
int zlib_inflate_padded(const uint8_t *src, size_t n, uint8_t *dst, size_t cap, size_t *produced) {
  if (n < 6) return fail(CASMVS_IO_CORRUPT, "zlib stream: %zu bytes is too short", n);
  const unsigned cmf = src[0], flg = src[1];
  if ((cmf & 15) != 8 || (cmf >> 4) > 7 || ((cmf << 8) | flg) % 31 != 0) return fail(CASMVS_IO_CORRUPT, "zlib stream: incorrect header check");
  if (flg & 0x20) return fail(CASMVS_IO_CORRUPT, "zlib stream: preset dictionary");
  std::unique_ptr<Inflater> inf(new Inflater);
  inf->in = src + 2;
  inf->in_end = src + n;
  inf->out = inf->out_begin = dst;
  inf->out_limit = dst + cap;
  if (int rc = inf->run()) return rc;
  if (inf->out > inf->out_limit) return fail(CASMVS_IO_CORRUPT, "inflate: more data than the image holds");
  if (inf->in + 4 > src + n) return fail(CASMVS_IO_CORRUPT, "zlib stream: missing check value");
  const size_t got = (size_t)(inf->out - dst);
  if (be32(inf->in) != adler32(dst, got)) return fail(CASMVS_IO_CORRUPT, "zlib stream: incorrect data check");
  *produced = got;
  return 0;
}

// ---- PNG ------------------------------------------------------------------------------------------------------------
uint32_t crc32_of(const uint8_t *p, size_t n) {   // only the small critical chunks are checked (IHDR, PLTE)
  static const uint32_t *table = [] {
    uint32_t *t = new uint32_t[256];
    for (uint32_t i = 0; i < 256; ++i) {
      uint32_t c = i;
      for (int k = 0; k < 8; ++k) c = (c & 1) ? 0xedb88320u ^ (c >> 1) : c >> 1;
      t[i] = c;
    }
    return t;
  }();
  uint32_t c = 0xffffffffu;
  for (size_t i = 0; i < n; ++i) c = table[(c ^ p[i]) & 255] ^ (c >> 8);
  return c ^ 0xffffffffu;
}

struct PngHeader {
  int32_t width = 0, height = 0, channels = 0, color_type = 0;
  const uint8_t *palette = nullptr;
  int palette_entries = 0;
  size_t idat_bytes = 0;
};

constexpr uint8_t kSignature[8] = {0x89, 'P', 'N', 'G', '\r', '\n', 0x1a, '\n'};

// Walks the chunks.  With `idat` non-null the IDAT payloads are concatenated into it (capacity from a first pass).
int parse_png(const uint8_t *file, size_t n, PngHeader *h, uint8_t *idat) {
  if (n < 8 + 25 || memcmp(file, kSignature, 8) != 0) return fail(CASMVS_IO_CORRUPT, "not a PNG file (bad signature)");
  size_t p = 8, copied = 0;
  bool have_ihdr = false, have_iend = false;
  h->idat_bytes = 0;
  while (p + 12 <= n) {
    const uint32_t len = be32(file + p);
    const uint8_t *type = file + p + 4, *data = file + p + 8;
    if (len > n - p - 12) return fail(CASMVS_IO_CORRUPT, "PNG chunk %.4s runs past the end of the file", (const char *)type);
    if (!have_ihdr) {
      if (memcmp(type, "IHDR", 4) != 0 || len != 13) return fail(CASMVS_IO_CORRUPT, "PNG: first chunk is not IHDR");
      if (crc32_of(type, 4 + 13) != be32(data + 13)) return fail(CASMVS_IO_CORRUPT, "PNG: IHDR checksum mismatch");
      h->width = (int32_t)be32(data);
      h->height = (int32_t)be32(data + 4);
      const int depth = data[8], ct = data[9];
      if (h->width <= 0 || h->height <= 0 || (int64_t)h->width * h->height > ((int64_t)1 << 28)) return fail(CASMVS_IO_CORRUPT, "PNG: bad image size %d x %d", h->width, h->height);
      if (data[10] != 0 || data[11] != 0 || data[12] > 1 || !(ct == 0 || ct == 2 || ct == 3 || ct == 4 || ct == 6) ||
          !(depth == 1 || depth == 2 || depth == 4 || depth == 8 || depth == 16))
        return fail(CASMVS_IO_CORRUPT, "PNG: invalid IHDR fields");
      if (depth != 8) return fail(CASMVS_IO_UNSUPPORTED, "PNG: %d-bit samples (8 only)", depth);
      if (data[12] != 0) return fail(CASMVS_IO_UNSUPPORTED, "PNG: interlaced");
      h->color_type = ct;
      h->channels = ct == 0 ? 1 : ct == 2 ? 3 : ct == 3 ? 1 : ct == 4 ? 2 : 4;
      have_ihdr = true;
    } else if (memcmp(type, "PLTE", 4) == 0) {
      if (len % 3 != 0 || len > 768 || len == 0) return fail(CASMVS_IO_CORRUPT, "PNG: bad PLTE length %u", len);
      if (crc32_of(type, 4 + len) != be32(data + len)) return fail(CASMVS_IO_CORRUPT, "PNG: PLTE checksum mismatch");
      h->palette = data;
      h->palette_entries = (int)(len / 3);
    } else if (memcmp(type, "IDAT", 4) == 0) {
      if (idat) memcpy(idat + copied, data, len);
      copied += len;
    } else if (memcmp(type, "IEND", 4) == 0) {
      have_iend = true;
      break;
    } else if (!(type[0] & 0x20)) {
      return fail(CASMVS_IO_UNSUPPORTED, "PNG: unknown critical chunk %.4s", (const char *)type);
    }
    p += 12 + (size_t)len;
  }
  if (!have_ihdr) return fail(CASMVS_IO_CORRUPT, "PNG: no IHDR");
  if (copied == 0) return fail(CASMVS_IO_CORRUPT, "PNG: no image data%s", have_iend ? "" : " (truncated file)");
  if (h->color_type == 3 && !h->palette) return fail(CASMVS_IO_CORRUPT, "PNG: palette image without PLTE");
  h->idat_bytes = copied;
  return 0;
}

inline int paeth(int a, int b, int c) {
  const int p = b - c, q = a - c;
  const int pa = p < 0 ? -p : p, pb = q < 0 ? -q : q, pc = (p + q) < 0 ? -(p + q) : (p + q);
  return (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
}

// dst = reconstruction of the filtered row `src` (ISO 15948 clause 9); prev = the reconstructed row above (zeros for row 0)
template <int BPP>
void unfilter_row(int filter, const uint8_t *src, const uint8_t *prev, uint8_t *dst, size_t n) {
  switch (filter) {
    case 0: memcpy(dst, src, n); break;
    case 1:
      for (size_t i = 0; i < (size_t)BPP && i < n; ++i) dst[i] = src[i];
      for (size_t i = BPP; i < n; ++i) dst[i] = (uint8_t)(src[i] + dst[i - BPP]);
      break;
    case 2:
      for (size_t i = 0; i < n; ++i) dst[i] = (uint8_t)(src[i] + prev[i]);
      break;
    case 3:
      for (size_t i = 0; i < (size_t)BPP && i < n; ++i) dst[i] = (uint8_t)(src[i] + (prev[i] >> 1));
      for (size_t i = BPP; i < n; ++i) dst[i] = (uint8_t)(src[i] + ((dst[i - BPP] + prev[i]) >> 1));
      break;
    default:
      for (size_t i = 0; i < (size_t)BPP && i < n; ++i) dst[i] = (uint8_t)(src[i] + prev[i]);
      for (size_t i = BPP; i < n; ++i) dst[i] = (uint8_t)(src[i] + paeth(dst[i - BPP], prev[i], prev[i - BPP]));
      break;
  }
}

void unfilter_any(int bpp, int filter, const uint8_t *src, const uint8_t *prev, uint8_t *dst, size_t n) {
  switch (bpp) {
    case 1: unfilter_row<1>(filter, src, prev, dst, n); break;
    case 2: unfilter_row<2>(filter, src, prev, dst, n); break;
    case 3: unfilter_row<3>(filter, src, prev, dst, n); break;
    default: unfilter_row<4>(filter, src, prev, dst, n); break;
  }
}

inline uint8_t luma601(unsigned r, unsigned g, unsigned b) { return (uint8_t)((r * 19595u + g * 38470u + b * 7471u + 0x8000u) >> 16); }   // PIL's L24 >> 16

// One reconstructed row of stored pixels -> one output row (what PIL's convert("RGB") / convert("L") produce)
void convert_row(const PngHeader &h, const uint8_t *row, uint8_t *out, int out_channels) {
  const int w = h.width;
  uint8_t pal_rgb[3] = {0, 0, 0};
  switch (h.color_type) {
    case 0:
      if (out_channels == 1) memcpy(out, row, (size_t)w);
      else for (int x = 0; x < w; ++x) out[3 * x] = out[3 * x + 1] = out[3 * x + 2] = row[x];
      break;
    case 4:
      if (out_channels == 1) for (int x = 0; x < w; ++x) out[x] = row[2 * x];
      else for (int x = 0; x < w; ++x) out[3 * x] = out[3 * x + 1] = out[3 * x + 2] = row[2 * x];
      break;
    case 2:
      if (out_channels == 3) memcpy(out, row, (size_t)w * 3);
      else for (int x = 0; x < w; ++x) out[x] = luma601(row[3 * x], row[3 * x + 1], row[3 * x + 2]);
      break;
    case 6:
      if (out_channels == 3) for (int x = 0; x < w; ++x) { out[3 * x] = row[4 * x]; out[3 * x + 1] = row[4 * x + 1]; out[3 * x + 2] = row[4 * x + 2]; }
      else for (int x = 0; x < w; ++x) out[x] = luma601(row[4 * x], row[4 * x + 1], row[4 * x + 2]);
      break;
    default:   // palette: indices beyond the PLTE entries read as black (PIL pads its palette with zeros)
      for (int x = 0; x < w; ++x) {
        const uint8_t *c = row[x] < h.palette_entries ? h.palette + 3 * row[x] : pal_rgb;
        if (out_channels == 3) { out[3 * x] = c[0]; out[3 * x + 1] = c[1]; out[3 * x + 2] = c[2]; }
        else out[x] = luma601(c[0], c[1], c[2]);
      }
      break;
  }
}

int decode_png(const uint8_t *file, size_t n, uint8_t *out, size_t out_row_bytes, int out_channels, int want_w, int want_h) {
  if (!file || !out || !(out_channels == 1 || out_channels == 3)) return fail(CASMVS_IO_BAD_ARGUMENT, "png_decode: null pointer or out_channels not 1 / 3");
  PngHeader h;
  if (int rc = parse_png(file, n, &h, nullptr)) return rc;
  if (want_w > 0 && (h.width != want_w || h.height != want_h)) return fail(CASMVS_IO_BAD_ARGUMENT, "png_decode: the file holds %d x %d pixels, the caller expects %d x %d", h.width, h.height, want_w, want_h);
  if (out_row_bytes < (size_t)h.width * out_channels) return fail(CASMVS_IO_BAD_ARGUMENT, "png_decode: out_row_bytes %zu < %d pixels x %d", out_row_bytes, h.width, out_channels);
  const size_t row = (size_t)h.width * h.channels, raw_bytes = (row + 1) * (size_t)h.height;
  // one allocation: [compressed stream + padding][filtered scanlines + slack][two scratch rows]
  const size_t a_off = 0, b_off = (h.idat_bytes + IN_PAD + 63) & ~(size_t)63, c_off = b_off + ((raw_bytes + OUT_SLACK + 63) & ~(size_t)63);
  uint8_t *buf = (uint8_t *)malloc(c_off + 2 * row + 64);
  if (!buf) return fail(CASMVS_IO_BAD_ARGUMENT, "png_decode: out of memory");
  std::unique_ptr<uint8_t, void (*)(void *)> guard(buf, free);
  uint8_t *idat = buf + a_off, *raw = buf + b_off, *scratch = buf + c_off;
  if (int rc = parse_png(file, n, &h, idat)) return rc;
  memset(idat + h.idat_bytes, 0, IN_PAD);
  size_t produced = 0;
  if (int rc = zlib_inflate_padded(idat, h.idat_bytes, raw, raw_bytes, &produced)) return rc;
  if (produced != raw_bytes) return fail(CASMVS_IO_CORRUPT, "PNG: %zu bytes of scanlines, %zu expected", produced, raw_bytes);
  const bool direct = (h.color_type == 2 && out_channels == 3) || (h.color_type == 0 && out_channels == 1);
  memset(scratch, 0, 2 * row);
  const uint8_t *prev = scratch;   // zeros above row 0
  for (int y = 0; y < h.height; ++y) {
    const uint8_t *src = raw + (size_t)y * (row + 1);
    const int filter = src[0];
    if (filter > 4) return fail(CASMVS_IO_CORRUPT, "PNG: filter type %d in row %d", filter, y);
    uint8_t *dst = direct ? out + (size_t)y * out_row_bytes : scratch + (size_t)((y & 1) ^ 1) * row;
    unfilter_any(h.channels, filter, src + 1, prev, dst, row);
    if (!direct) convert_row(h, dst, out + (size_t)y * out_row_bytes, out_channels);
    prev = dst;
  }
  return 0;
}

struct FileBytes {
  uint8_t *data = nullptr;
  size_t size = 0;
  ~FileBytes() { free(data); }
};

int read_whole_file(const char *path, FileBytes *f) {
  const int fd = open(path, O_RDONLY | O_CLOEXEC);
  if (fd < 0) return fail(CASMVS_IO_BAD_ARGUMENT, "cannot open %s", path);
  struct stat st;
  if (fstat(fd, &st) != 0 || st.st_size <= 0) {
    close(fd);
    return fail(CASMVS_IO_CORRUPT, "cannot stat %s (or it is empty)", path);
  }
  f->size = (size_t)st.st_size;
  f->data = (uint8_t *)malloc(f->size);
  size_t got = 0;
  while (f->data && got < f->size) {
    const ssize_t r = read(fd, f->data + got, f->size - got);
    if (r <= 0) break;
    got += (size_t)r;
  }
  close(fd);
  if (!f->data || got != f->size) return fail(CASMVS_IO_CORRUPT, "short read of %s", path);
  return 0;
}

}  // namespace

extern "C" {

const char *casmvs_io_last_error(void) { return g_error; }

int casmvs_png_info(const uint8_t *file, size_t file_bytes, int32_t *width, int32_t *height, int32_t *channels) {
  g_error[0] = 0;
  if (!file || !width || !height || !channels) return fail(CASMVS_IO_BAD_ARGUMENT, "png_info: null pointer");
  PngHeader h;
  if (int rc = parse_png(file, file_bytes, &h, nullptr)) return rc;
  *width = h.width;
  *height = h.height;
  *channels = h.color_type == 3 ? 3 : h.channels;
  return 0;
}

int casmvs_png_decode(const uint8_t *file, size_t file_bytes, uint8_t *out, size_t out_row_bytes, int32_t out_channels) {
  g_error[0] = 0;
  return decode_png(file, file_bytes, out, out_row_bytes, out_channels, 0, 0);
}

int casmvs_png_decode_file(const char *path, uint8_t *out, size_t out_row_bytes, int32_t width, int32_t height, int32_t out_channels) {
  g_error[0] = 0;
  if (!path || width <= 0 || height <= 0) return fail(CASMVS_IO_BAD_ARGUMENT, "png_decode_file: null path or bad size");
  FileBytes f;
  if (int rc = read_whole_file(path, &f)) return rc;
  return decode_png(f.data, f.size, out, out_row_bytes, out_channels, width, height);
}

int casmvs_png_decode_files(const char *const *paths, int32_t n, uint8_t *out, size_t image_stride, size_t out_row_bytes, int32_t width,
                            int32_t height, int32_t out_channels, int32_t threads, int32_t *status) {
  g_error[0] = 0;
  if (n < 0 || (n > 0 && (!paths || !out || !status))) return fail(CASMVS_IO_BAD_ARGUMENT, "png_decode_files: null pointer");
  if (n > 0 && image_stride < out_row_bytes * (size_t)height) return fail(CASMVS_IO_BAD_ARGUMENT, "png_decode_files: image_stride smaller than one image");
  if (n == 0) return 0;
  int hw = (int)std::thread::hardware_concurrency();
  if (hw <= 0) hw = 1;
  int nt = threads > 0 ? threads : hw;
  if (nt > n) nt = n;
  std::atomic<int> next{0};
  std::vector<std::string> messages((size_t)n);
  auto work = [&] {
    for (int i = next.fetch_add(1); i < n; i = next.fetch_add(1)) {
      status[i] = casmvs_png_decode_file(paths[i], out + (size_t)i * image_stride, out_row_bytes, width, height, out_channels);
      if (status[i]) messages[(size_t)i] = g_error;   // the worker thread's message
    }
  };
  std::vector<std::thread> pool;
  for (int t = 1; t < nt; ++t) pool.emplace_back(work);
  work();
  for (auto &t : pool) t.join();
  for (int i = 0; i < n; ++i)
    if (status[i]) return fail(status[i], "%s: %s", paths[i], messages[(size_t)i].c_str());
  g_error[0] = 0;
  return 0;
}

int casmvs_zlib_inflate(const uint8_t *src, size_t src_bytes, uint8_t *dst, size_t dst_capacity, size_t *out_bytes) {
  g_error[0] = 0;
  if (!src || !out_bytes || (!dst && dst_capacity)) return fail(CASMVS_IO_BAD_ARGUMENT, "zlib_inflate: null pointer");
  const uint16_t probe = 1;
  if (*(const uint8_t *)&probe != 1) return fail(CASMVS_IO_UNSUPPORTED, "big-endian host");
  std::vector<uint8_t> in(src_bytes + IN_PAD, 0), outbuf(dst_capacity + OUT_SLACK);
  memcpy(in.data(), src, src_bytes);
  size_t produced = 0;
  if (int rc = zlib_inflate_padded(in.data(), src_bytes, outbuf.data(), dst_capacity, &produced)) return rc;
  memcpy(dst, outbuf.data(), produced);
  *out_bytes = produced;
  return 0;
}

}  // extern "C"
