/*
 * ORACLE (test infrastructure only -- never linked or imported by the product path).
 *
 * CPU JPEG decoder restating what the reference's `decoders.image(device="cpu")` computes.
 *
 * In the reference the arithmetic is NOT in tree: ImageDecoder hands the batch to nvImageCodec
 * (dali/operators/imgcodec/image_decoder.h:265,278,289-321,380,482-483,810-815), pinned to
 * nvImageCodec v0.9.0 (cmake/Dependencies.common.cmake:310-311,378-379), whose CPU JPEG backend is
 * the `libjpeg_turbo_decoder` extension over libjpeg-turbo, configured as: accurate integer IDCT
 * (JDCT_ISLOW) unless use_fast_idct (image_decoder.h:290-291), fancy upsampling always on for the
 * CPU path (image_decoder.h:297-305), output I_RGB interleaved u8 (image_decoder.h:524-607).
 * This file therefore restates libjpeg-turbo's published algorithms:
 *   ITU-T T.81 Huffman entropy decoding, sequential (jdhuff.c) and progressive (jdphuff.c);
 *   jidctint.c  jpeg_idct_islow  (CONST_BITS 13, PASS1_BITS 2);
 *   jdsample.c  h2v1_fancy / h2v2_fancy / h1v2_fancy (triangle) and integral box upsampling,
 *               with jdmainct.c's edge-row replication for the context rows;
 *   jdcolor.c   YCbCr -> RGB with 16-bit fixed-point tables (BT.601 full range).
 *
 * Pinning: the reference holds no golden pixels for lossy JPEG (only cpu-vs-mixed tolerances,
 * dali/test/python/decoder/test_image.py:221-323).  This oracle is pinned BIT-EXACT against
 * libjpeg-turbo 3.1.4.1 as bundled in Pillow 12.2.0 (Image.open(..).convert("RGB")) over
 * synthetic baseline/progressive 4:4:4 / 4:2:2 / 4:2:0 / 4:4:0 / grayscale streams, odd sizes and
 * restart intervals: tests/test_oracle_jpeg.py, with fixtures under tests/golden/.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define MAX_COMPS 4

typedef struct {
  uint8_t bits[17];
  uint8_t vals[256];
  int present;
  /* derived */
  int32_t maxcode[18];
  int32_t valoffset[17];
  uint8_t look_nbits[256];
  uint8_t look_sym[256];
} huff_tbl;

typedef struct {
  int id, h, v, tq;
  int wblk, hblk;      /* allocated blocks (padded to the interleaved MCU) */
  int dw, dh;          /* downsampled_width / downsampled_height (samples) */
  int16_t *coef;       /* [hblk][wblk][64], natural order */
  uint8_t *plane;      /* [hblk*8][wblk*8] */
} comp_t;

typedef struct {
  const uint8_t *data;
  size_t size, pos;
  /* bit reader */
  uint64_t bitbuf;
  int bitcnt;
  int hit_marker;
  /* frame */
  int width, height, ncomp, progressive, precision;
  int hmax, vmax;
  comp_t comp[MAX_COMPS];
  uint16_t qt[4][64]; /* natural order */
  int qt_present[4];
  huff_tbl dc[4], ac[4];
  int restart_interval;
  int saw_jfif, saw_adobe, adobe_transform;
  int orientation;
  /* scan */
  int scan_ncomp, scan_comp[MAX_COMPS], scan_td[MAX_COMPS], scan_ta[MAX_COMPS];
  int Ss, Se, Ah, Al;
  int last_dc[MAX_COMPS];
  int eobrun;
} dec_t;

static const uint8_t zigzag_natural[64 + 16] = {
    0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
    41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
    30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63,
    63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63};

/* ------------------------------------------------------------------ Huffman tables */
static int build_huff(huff_tbl *t) {
  int code = 0, p = 0;
  char huffsize[257];
  unsigned int huffcode[257];
  for (int l = 1; l <= 16; l++) {
    int n = t->bits[l];
    if (p + n > 256) return 1;
    while (n--) huffsize[p++] = (char)l;
  }
  huffsize[p] = 0;
  int numsymbols = p;
  int si = huffsize[0];
  p = 0;
  while (huffsize[p]) {
    while (((int)huffsize[p]) == si) { huffcode[p++] = code; code++; }
    if (code >= (1 << si)) return 1;
    code <<= 1;
    si++;
  }
  p = 0;
  for (int l = 1; l <= 16; l++) {
    if (t->bits[l]) {
      t->valoffset[l] = p - (int)huffcode[p];
      p += t->bits[l];
      t->maxcode[l] = huffcode[p - 1];
    } else {
      t->maxcode[l] = -1;
    }
  }
  t->maxcode[17] = 0xFFFFF;
  memset(t->look_nbits, 0, sizeof(t->look_nbits));
  p = 0;
  for (int l = 1; l <= 8; l++) {
    for (int i = 1; i <= t->bits[l]; i++, p++) {
      int lookbits = huffcode[p] << (8 - l);
      for (int ctr = 1 << (8 - l); ctr > 0; ctr--) {
        t->look_nbits[lookbits] = (uint8_t)l;
        t->look_sym[lookbits] = t->vals[p];
        lookbits++;
      }
    }
  }
  (void)numsymbols;
  t->present = 1;
  return 0;
}

/* ------------------------------------------------------------------ bit reader */
static void fill_bits(dec_t *d) {
  while (d->bitcnt <= 56) {
    int c = 0;
    if (!d->hit_marker && d->pos < d->size) {
      c = d->data[d->pos];
      if (c == 0xFF) {
        int c2 = d->pos + 1 < d->size ? d->data[d->pos + 1] : 0xD9;
        if (c2 == 0) {
          d->pos += 2;
        } else {
          d->hit_marker = 1; /* leave pos at the marker; feed zeros (jdhuff.c behaviour) */
          c = 0;
        }
      } else {
        d->pos++;
      }
    } else {
      d->hit_marker = 1;
    }
    d->bitbuf |= (uint64_t)c << (56 - d->bitcnt);
    d->bitcnt += 8;
  }
}
static inline int peek_bits(dec_t *d, int n) {
  if (d->bitcnt < n) fill_bits(d);
  return (int)(d->bitbuf >> (64 - n));
}
static inline void drop_bits(dec_t *d, int n) { d->bitbuf <<= n; d->bitcnt -= n; }
static inline int get_bits(dec_t *d, int n) {
  if (n == 0) return 0;
  int v = peek_bits(d, n);
  drop_bits(d, n);
  return v;
}
static inline int huff_extend(int x, int s) { return x < (1 << (s - 1)) ? x + (int)((~0u) << s) + 1 : x; }

static int huff_decode(dec_t *d, const huff_tbl *t) {
  int look = peek_bits(d, 8);
  int nb = t->look_nbits[look];
  if (nb) { drop_bits(d, nb); return t->look_sym[look]; }
  int l = 9;
  int code = peek_bits(d, 16);
  for (; l <= 16; l++) {
    int c = code >> (16 - l);
    if (c <= t->maxcode[l]) {
      drop_bits(d, l);
      return t->vals[(c + t->valoffset[l]) & 0xFF];
    }
  }
  drop_bits(d, 16);
  return 0; /* corrupt: jdhuff.c returns 0 with a warning */
}

/* ------------------------------------------------------------------ markers */
static int rd16(const uint8_t *p) { return (p[0] << 8) | p[1]; }

static void parse_exif(dec_t *d, const uint8_t *p, int len) {
  if (len < 14 || memcmp(p, "Exif\0\0", 6)) return;
  const uint8_t *t = p + 6;
  int n = len - 6;
  int le;
  if (t[0] == 'I' && t[1] == 'I') le = 1; else if (t[0] == 'M' && t[1] == 'M') le = 0; else return;
#define R16(o) (le ? (t[o] | (t[(o) + 1] << 8)) : ((t[o] << 8) | t[(o) + 1]))
#define R32(o) (le ? ((uint32_t)t[o] | ((uint32_t)t[(o) + 1] << 8) | ((uint32_t)t[(o) + 2] << 16) | ((uint32_t)t[(o) + 3] << 24)) \
                   : (((uint32_t)t[o] << 24) | ((uint32_t)t[(o) + 1] << 16) | ((uint32_t)t[(o) + 2] << 8) | t[(o) + 3]))
  if (R16(2) != 42) return;
  uint32_t off = R32(4);
  if (off + 2 > (uint32_t)n) return;
  int cnt = R16(off);
  for (int i = 0; i < cnt; i++) {
    uint32_t e = off + 2 + 12 * i;
    if (e + 12 > (uint32_t)n) return;
    if (R16(e) == 0x0112) {
      int v = R16(e + 8);
      if (v >= 1 && v <= 8) d->orientation = v;
      return;
    }
  }
#undef R16
#undef R32
}

static int parse_sof(dec_t *d, const uint8_t *p, int len, int progressive) {
  if (len < 6) return 1;
  d->precision = p[0];
  d->height = rd16(p + 1);
  d->width = rd16(p + 3);
  d->ncomp = p[5];
  d->progressive = progressive;
  if (d->precision != 8 || d->ncomp < 1 || d->ncomp > MAX_COMPS || len < 6 + 3 * d->ncomp) return 1;
  if (d->width <= 0 || d->height <= 0) return 1;
  d->hmax = d->vmax = 1;
  for (int i = 0; i < d->ncomp; i++) {
    comp_t *c = &d->comp[i];
    c->id = p[6 + 3 * i];
    c->h = p[7 + 3 * i] >> 4;
    c->v = p[7 + 3 * i] & 15;
    c->tq = p[8 + 3 * i] & 3;
    if (c->h < 1 || c->h > 4 || c->v < 1 || c->v > 4) return 1;
    if (c->h > d->hmax) d->hmax = c->h;
    if (c->v > d->vmax) d->vmax = c->v;
  }
  int mcux = (d->width + 8 * d->hmax - 1) / (8 * d->hmax);
  int mcuy = (d->height + 8 * d->vmax - 1) / (8 * d->vmax);
  for (int i = 0; i < d->ncomp; i++) {
    comp_t *c = &d->comp[i];
    c->wblk = mcux * c->h;
    c->hblk = mcuy * c->v;
    c->dw = (d->width * c->h + d->hmax - 1) / d->hmax;
    c->dh = (d->height * c->v + d->vmax - 1) / d->vmax;
    c->coef = (int16_t *)calloc((size_t)c->wblk * c->hblk * 64, sizeof(int16_t));
    c->plane = (uint8_t *)malloc((size_t)c->wblk * c->hblk * 64);
    if (!c->coef || !c->plane) return 1;
  }
  return 0;
}

static int parse_dqt(dec_t *d, const uint8_t *p, int len) {
  while (len > 0) {
    int pq = p[0] >> 4, tq = p[0] & 15;
    if (tq > 3) return 1;
    p++; len--;
    for (int i = 0; i < 64; i++) {
      int v;
      if (pq) { if (len < 2) return 1; v = rd16(p); p += 2; len -= 2; }
      else { if (len < 1) return 1; v = p[0]; p++; len--; }
      d->qt[tq][zigzag_natural[i]] = (uint16_t)v;
    }
    d->qt_present[tq] = 1;
  }
  return 0;
}

static int parse_dht(dec_t *d, const uint8_t *p, int len) {
  while (len > 0) {
    if (len < 17) return 1;
    int tc = p[0] >> 4, th = p[0] & 15;
    if (th > 3 || tc > 1) return 1;
    huff_tbl *t = tc ? &d->ac[th] : &d->dc[th];
    int count = 0;
    t->bits[0] = 0;
    for (int i = 1; i <= 16; i++) { t->bits[i] = p[i]; count += p[i]; }
    p += 17; len -= 17;
    if (count > 256 || count > len) return 1;
    memset(t->vals, 0, sizeof(t->vals));
    memcpy(t->vals, p, (size_t)count);
    p += count; len -= count;
    if (build_huff(t)) return 1;
  }
  return 0;
}

/* ------------------------------------------------------------------ entropy decoding */
static void reset_scan_state(dec_t *d) {
  memset(d->last_dc, 0, sizeof(d->last_dc));
  d->eobrun = 0;
}

/* Called at a restart boundary: discard bits, consume RSTn if present. */
static void process_restart(dec_t *d) {
  d->bitbuf = 0;
  d->bitcnt = 0;
  /* scan forward to the marker */
  while (d->pos + 1 < d->size) {
    if (d->data[d->pos] == 0xFF && d->data[d->pos + 1] >= 0xD0 && d->data[d->pos + 1] <= 0xD7) {
      d->pos += 2;
      break;
    }
    if (d->data[d->pos] == 0xFF && d->data[d->pos + 1] != 0 && d->data[d->pos + 1] != 0xFF) break;
    d->pos++;
  }
  d->hit_marker = 0;
  reset_scan_state(d);
}

static void decode_block_seq(dec_t *d, int16_t *blk, int ci_scan) {
  int ci = d->scan_comp[ci_scan];
  const huff_tbl *dct = &d->dc[d->scan_td[ci_scan]];
  const huff_tbl *act = &d->ac[d->scan_ta[ci_scan]];
  int s = huff_decode(d, dct);
  if (s) { int r = get_bits(d, s); s = huff_extend(r, s); }
  d->last_dc[ci] += s;
  blk[0] = (int16_t)d->last_dc[ci];
  for (int k = 1; k < 64; k++) {
    s = huff_decode(d, act);
    int r = s >> 4;
    s &= 15;
    if (s) {
      k += r;
      r = get_bits(d, s);
      s = huff_extend(r, s);
      blk[zigzag_natural[k]] = (int16_t)s;
    } else {
      if (r != 15) break;
      k += 15;
    }
  }
}

static void decode_block_dc_first(dec_t *d, int16_t *blk, int ci_scan) {
  int ci = d->scan_comp[ci_scan];
  int s = huff_decode(d, &d->dc[d->scan_td[ci_scan]]);
  if (s) { int r = get_bits(d, s); s = huff_extend(r, s); }
  d->last_dc[ci] += s;
  blk[0] = (int16_t)(d->last_dc[ci] * (1 << d->Al));
}
static void decode_block_dc_refine(dec_t *d, int16_t *blk) {
  if (get_bits(d, 1)) blk[0] |= (int16_t)(1 << d->Al);
}
static void decode_block_ac_first(dec_t *d, int16_t *blk) {
  const huff_tbl *t = &d->ac[d->scan_ta[0]];
  if (d->eobrun > 0) { d->eobrun--; return; }
  for (int k = d->Ss; k <= d->Se; k++) {
    int s = huff_decode(d, t);
    int r = s >> 4;
    s &= 15;
    if (s) {
      k += r;
      r = get_bits(d, s);
      s = huff_extend(r, s);
      blk[zigzag_natural[k]] = (int16_t)(s * (1 << d->Al));
    } else {
      if (r == 15) { k += 15; }
      else {
        d->eobrun = 1 << r;
        if (r) d->eobrun += get_bits(d, r);
        d->eobrun--;
        break;
      }
    }
  }
}
static void decode_block_ac_refine(dec_t *d, int16_t *blk) {
  const huff_tbl *t = &d->ac[d->scan_ta[0]];
  int p1 = 1 << d->Al, m1 = (int)((~0u) << d->Al);
  int k = d->Ss;
  if (d->eobrun == 0) {
    for (; k <= d->Se; k++) {
      int s = huff_decode(d, t);
      int r = s >> 4;
      s &= 15;
      if (s) {
        s = get_bits(d, 1) ? p1 : m1;
      } else {
        if (r != 15) {
          d->eobrun = 1 << r;
          if (r) d->eobrun += get_bits(d, r);
          break;
        }
      }
      do {
        int16_t *c = blk + zigzag_natural[k];
        if (*c != 0) {
          if (get_bits(d, 1)) {
            if ((*c & p1) == 0) { if (*c >= 0) *c = (int16_t)(*c + p1); else *c = (int16_t)(*c + m1); }
          }
        } else {
          if (--r < 0) break;
        }
        k++;
      } while (k <= d->Se);
      if (s) blk[zigzag_natural[k]] = (int16_t)s;
    }
  }
  if (d->eobrun > 0) {
    for (; k <= d->Se; k++) {
      int16_t *c = blk + zigzag_natural[k];
      if (*c != 0) {
        if (get_bits(d, 1)) {
          if ((*c & p1) == 0) { if (*c >= 0) *c = (int16_t)(*c + p1); else *c = (int16_t)(*c + m1); }
        }
      }
    }
    d->eobrun--;
  }
}

static void decode_one_block(dec_t *d, int16_t *blk, int ci_scan) {
  if (!d->progressive) { decode_block_seq(d, blk, ci_scan); return; }
  if (d->Ss == 0) {
    if (d->Ah == 0) decode_block_dc_first(d, blk, ci_scan); else decode_block_dc_refine(d, blk);
  } else {
    if (d->Ah == 0) decode_block_ac_first(d, blk); else decode_block_ac_refine(d, blk);
  }
}

static int decode_scan(dec_t *d) {
  d->bitbuf = 0; d->bitcnt = 0; d->hit_marker = 0;
  reset_scan_state(d);
  int restarts_left = d->restart_interval;
  if (d->scan_ncomp == 1) {
    comp_t *c = &d->comp[d->scan_comp[0]];
    int bw = (c->dw + 7) / 8, bh = (c->dh + 7) / 8;
    for (int by = 0; by < bh; by++)
      for (int bx = 0; bx < bw; bx++) {
        if (d->restart_interval) {
          if (restarts_left == 0) { process_restart(d); restarts_left = d->restart_interval; }
          restarts_left--;
        }
        decode_one_block(d, c->coef + ((size_t)by * c->wblk + bx) * 64, 0);
      }
  } else {
    int mcux = (d->width + 8 * d->hmax - 1) / (8 * d->hmax);
    int mcuy = (d->height + 8 * d->vmax - 1) / (8 * d->vmax);
    for (int my = 0; my < mcuy; my++)
      for (int mx = 0; mx < mcux; mx++) {
        if (d->restart_interval) {
          if (restarts_left == 0) { process_restart(d); restarts_left = d->restart_interval; }
          restarts_left--;
        }
        for (int s = 0; s < d->scan_ncomp; s++) {
          comp_t *c = &d->comp[d->scan_comp[s]];
          for (int v = 0; v < c->v; v++)
            for (int h = 0; h < c->h; h++) {
              int by = my * c->v + v, bx = mx * c->h + h;
              decode_one_block(d, c->coef + ((size_t)by * c->wblk + bx) * 64, s);
            }
        }
      }
  }
  /* leave pos at the next marker: skip any residual entropy bytes */
  while (d->pos + 1 < d->size) {
    if (d->data[d->pos] == 0xFF && d->data[d->pos + 1] != 0 &&
        !(d->data[d->pos + 1] >= 0xD0 && d->data[d->pos + 1] <= 0xD7) && d->data[d->pos + 1] != 0xFF)
      break;
    d->pos++;
  }
  return 0;
}

/* ------------------------------------------------------------------ IDCT (jidctint.c islow) */
#define CONST_BITS 13
#define PASS1_BITS 2
#define FIX_0_298631336 2446
#define FIX_0_390180644 3196
#define FIX_0_541196100 4433
#define FIX_0_765366865 6270
#define FIX_0_899976223 7373
#define FIX_1_175875602 9633
#define FIX_1_501321110 12299
#define FIX_1_847759065 15137
#define FIX_1_961570560 16069
#define FIX_2_053119869 16819
#define FIX_2_562915447 20995
#define FIX_3_072711026 25172
#define DESCALE(x, n) (((x) + (1 << ((n)-1))) >> (n))

static inline uint8_t idct_range_limit(int32_t x) {
  /* range_limit[(x) & RANGE_MASK] with the table centred on CENTERJSAMPLE (jdmaster.c
   * prepare_range_limit_table): signed 10-bit wrap, +128, clamp */
  int32_t v = ((x & 1023) ^ 512) - 512 + 128;
  return (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v);
}

void orc_idct_islow(const int16_t *coef, const uint16_t *q, uint8_t *out, int stride) {
  int32_t ws[64];
  for (int c = 0; c < 8; c++) {
    const int16_t *in = coef + c;
    const uint16_t *qq = q + c;
    int32_t *w = ws + c;
#define DQ(k) ((int32_t)in[8 * (k)] * (int32_t)qq[8 * (k)])
    if (in[8] == 0 && in[16] == 0 && in[24] == 0 && in[32] == 0 && in[40] == 0 && in[48] == 0 && in[56] == 0) {
      int32_t dc = DQ(0) * (1 << PASS1_BITS);
      for (int k = 0; k < 8; k++) w[8 * k] = dc;
      continue;
    }
    int32_t z2 = DQ(2), z3 = DQ(6);
    int32_t z1 = (z2 + z3) * FIX_0_541196100;
    int32_t tmp2 = z1 + z3 * (-FIX_1_847759065);
    int32_t tmp3 = z1 + z2 * FIX_0_765366865;
    z2 = DQ(0); z3 = DQ(4);
    int32_t tmp0 = (z2 + z3) * (1 << CONST_BITS);
    int32_t tmp1 = (z2 - z3) * (1 << CONST_BITS);
    int32_t tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
    tmp0 = DQ(7); tmp1 = DQ(5); tmp2 = DQ(3); tmp3 = DQ(1);
    z1 = tmp0 + tmp3; z2 = tmp1 + tmp2; z3 = tmp0 + tmp2;
    int32_t z4 = tmp1 + tmp3;
    int32_t z5 = (z3 + z4) * FIX_1_175875602;
    tmp0 *= FIX_0_298631336; tmp1 *= FIX_2_053119869; tmp2 *= FIX_3_072711026; tmp3 *= FIX_1_501321110;
    z1 *= -FIX_0_899976223; z2 *= -FIX_2_562915447; z3 *= -FIX_1_961570560; z4 *= -FIX_0_390180644;
    z3 += z5; z4 += z5;
    tmp0 += z1 + z3; tmp1 += z2 + z4; tmp2 += z2 + z3; tmp3 += z1 + z4;
    w[0] = DESCALE(tmp10 + tmp3, CONST_BITS - PASS1_BITS);
    w[56] = DESCALE(tmp10 - tmp3, CONST_BITS - PASS1_BITS);
    w[8] = DESCALE(tmp11 + tmp2, CONST_BITS - PASS1_BITS);
    w[48] = DESCALE(tmp11 - tmp2, CONST_BITS - PASS1_BITS);
    w[16] = DESCALE(tmp12 + tmp1, CONST_BITS - PASS1_BITS);
    w[40] = DESCALE(tmp12 - tmp1, CONST_BITS - PASS1_BITS);
    w[24] = DESCALE(tmp13 + tmp0, CONST_BITS - PASS1_BITS);
    w[32] = DESCALE(tmp13 - tmp0, CONST_BITS - PASS1_BITS);
#undef DQ
  }
  for (int r = 0; r < 8; r++) {
    const int32_t *w = ws + 8 * r;
    uint8_t *o = out + (size_t)r * stride;
    int32_t z2 = w[2], z3 = w[6];
    int32_t z1 = (z2 + z3) * FIX_0_541196100;
    int32_t tmp2 = z1 + z3 * (-FIX_1_847759065);
    int32_t tmp3 = z1 + z2 * FIX_0_765366865;
    int32_t tmp0 = (w[0] + w[4]) * (1 << CONST_BITS);
    int32_t tmp1 = (w[0] - w[4]) * (1 << CONST_BITS);
    int32_t tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
    tmp0 = w[7]; tmp1 = w[5]; tmp2 = w[3]; tmp3 = w[1];
    z1 = tmp0 + tmp3; z2 = tmp1 + tmp2; z3 = tmp0 + tmp2;
    int32_t z4 = tmp1 + tmp3;
    int32_t z5 = (z3 + z4) * FIX_1_175875602;
    tmp0 *= FIX_0_298631336; tmp1 *= FIX_2_053119869; tmp2 *= FIX_3_072711026; tmp3 *= FIX_1_501321110;
    z1 *= -FIX_0_899976223; z2 *= -FIX_2_562915447; z3 *= -FIX_1_961570560; z4 *= -FIX_0_390180644;
    z3 += z5; z4 += z5;
    tmp0 += z1 + z3; tmp1 += z2 + z4; tmp2 += z2 + z3; tmp3 += z1 + z4;
    const int S = CONST_BITS + PASS1_BITS + 3;
    o[0] = idct_range_limit(DESCALE(tmp10 + tmp3, S));
    o[7] = idct_range_limit(DESCALE(tmp10 - tmp3, S));
    o[1] = idct_range_limit(DESCALE(tmp11 + tmp2, S));
    o[6] = idct_range_limit(DESCALE(tmp11 - tmp2, S));
    o[2] = idct_range_limit(DESCALE(tmp12 + tmp1, S));
    o[5] = idct_range_limit(DESCALE(tmp12 - tmp1, S));
    o[3] = idct_range_limit(DESCALE(tmp13 + tmp0, S));
    o[4] = idct_range_limit(DESCALE(tmp13 - tmp0, S));
  }
}

/* ------------------------------------------------------------------ upsampling (jdsample.c) */
static inline int clampi(int v, int lo, int hi) { return v < lo ? lo : v > hi ? hi : v; }

/* Produces the full-resolution plane `dst` [H][W] for component c. */
static void upsample_component(const dec_t *d, const comp_t *c, uint8_t *dst) {
  int W = d->width, H = d->height;
  int stride = c->wblk * 8;
  int hx = d->hmax / c->h, vx = d->vmax / c->v; /* expansion factors when integral */
  int integral = (d->hmax % c->h == 0) && (d->vmax % c->v == 0);
  const uint8_t *src = c->plane;
  if (c->h == d->hmax && c->v == d->vmax) {
    for (int y = 0; y < H; y++) memcpy(dst + (size_t)y * W, src + (size_t)y * stride, (size_t)W);
    return;
  }
  int fancy_h2v1 = c->h * 2 == d->hmax && c->v == d->vmax && c->dw > 2;
  int fancy_h1v2 = c->h == d->hmax && c->v * 2 == d->vmax;
  int fancy_h2v2 = c->h * 2 == d->hmax && c->v * 2 == d->vmax && c->dw > 2;
  if (fancy_h2v1) {
    for (int y = 0; y < H; y++) {
      const uint8_t *in = src + (size_t)y * stride;
      uint8_t *o = dst + (size_t)y * W;
      int n = c->dw;
      for (int x = 0; x < W; x++) {
        int i = x >> 1;
        int v;
        if ((x & 1) == 0) v = i == 0 ? in[0] : (in[i] * 3 + in[i - 1] + 1) >> 2;
        else v = i == n - 1 ? in[i] : (in[i] * 3 + in[i + 1] + 2) >> 2;
        o[x] = (uint8_t)v;
      }
    }
  } else if (fancy_h1v2) {
    for (int y = 0; y < H; y++) {
      int i = y >> 1;
      int other = (y & 1) ? i + 1 : i - 1;
      other = clampi(other, 0, c->dh - 1);
      int bias = (y & 1) ? 2 : 1;
      const uint8_t *in0 = src + (size_t)i * stride, *in1 = src + (size_t)other * stride;
      uint8_t *o = dst + (size_t)y * W;
      for (int x = 0; x < W; x++) o[x] = (uint8_t)((in0[x] * 3 + in1[x] + bias) >> 2);
    }
  } else if (fancy_h2v2) {
    int n = c->dw;
    for (int y = 0; y < H; y++) {
      int i = y >> 1;
      int other = (y & 1) ? i + 1 : i - 1;
      other = clampi(other, 0, c->dh - 1);
      const uint8_t *in0 = src + (size_t)i * stride, *in1 = src + (size_t)other * stride;
      uint8_t *o = dst + (size_t)y * W;
      for (int x = 0; x < W; x++) {
        int k = x >> 1;
        int thiscol = in0[k] * 3 + in1[k];
        int v;
        if ((x & 1) == 0) {
          if (k == 0) v = (thiscol * 4 + 8) >> 4;
          else v = (thiscol * 3 + (in0[k - 1] * 3 + in1[k - 1]) + 8) >> 4;
        } else {
          if (k == n - 1) v = (thiscol * 4 + 7) >> 4;
          else v = (thiscol * 3 + (in0[k + 1] * 3 + in1[k + 1]) + 7) >> 4;
        }
        o[x] = (uint8_t)v;
      }
    }
  } else if (integral) {
    for (int y = 0; y < H; y++) {
      const uint8_t *in = src + (size_t)(y / vx) * stride;
      uint8_t *o = dst + (size_t)y * W;
      for (int x = 0; x < W; x++) o[x] = in[x / hx];
    }
  } else {
    memset(dst, 128, (size_t)W * H); /* fractional sampling: not supported by libjpeg either */
  }
}

/* ------------------------------------------------------------------ colour (jdcolor.c) */
#define SCALEBITS 16
#define ONE_HALF ((int32_t)1 << (SCALEBITS - 1))
#define FIXC(x) ((int32_t)((x) * (1L << SCALEBITS) + 0.5))
static inline uint8_t clamp8(int v) { return (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v); }

static void ycc_to_rgb(const uint8_t *y, const uint8_t *cb, const uint8_t *cr, uint8_t *rgb, size_t n) {
  static int32_t cr_r[256], cb_b[256], cr_g[256], cb_g[256];
  static int init = 0;
  if (!init) {
    for (int i = 0; i < 256; i++) {
      int32_t x = i - 128;
      cr_r[i] = (int32_t)(FIXC(1.40200) * x + ONE_HALF) >> SCALEBITS;
      cb_b[i] = (int32_t)(FIXC(1.77200) * x + ONE_HALF) >> SCALEBITS;
      cr_g[i] = (-FIXC(0.71414)) * x;
      cb_g[i] = (-FIXC(0.34414)) * x + ONE_HALF;
    }
    init = 1;
  }
  for (size_t i = 0; i < n; i++) {
    int yy = y[i];
    rgb[3 * i + 0] = clamp8(yy + cr_r[cr[i]]);
    rgb[3 * i + 1] = clamp8(yy + ((cb_g[cb[i]] + cr_g[cr[i]]) >> SCALEBITS));
    rgb[3 * i + 2] = clamp8(yy + cb_b[cb[i]]);
  }
}

/* ------------------------------------------------------------------ driver */
static void free_dec(dec_t *d) {
  for (int i = 0; i < MAX_COMPS; i++) { free(d->comp[i].coef); free(d->comp[i].plane); }
}

static int parse_until_eoi(dec_t *d, int headers_only) {
  if (d->size < 4 || d->data[0] != 0xFF || d->data[1] != 0xD8) return 1;
  d->pos = 2;
  d->orientation = 1;
  int have_frame = 0;
  for (;;) {
    /* find next marker */
    while (d->pos < d->size && d->data[d->pos] != 0xFF) d->pos++;
    while (d->pos < d->size && d->data[d->pos] == 0xFF) d->pos++;
    if (d->pos >= d->size) return have_frame ? 0 : 1;
    int m = d->data[d->pos++];
    if (m == 0xD9) return have_frame ? 0 : 1;
    if (m == 0x01 || (m >= 0xD0 && m <= 0xD7)) continue;
    if (d->pos + 2 > d->size) return have_frame ? 0 : 1;
    int len = rd16(d->data + d->pos);
    if (len < 2 || d->pos + len > d->size) return have_frame ? 0 : 1;
    const uint8_t *p = d->data + d->pos + 2;
    int plen = len - 2;
    d->pos += len;
    switch (m) {
      case 0xC0: case 0xC1:
        if (have_frame || parse_sof(d, p, plen, 0)) return 1;
        have_frame = 1;
        if (headers_only) return 0;
        break;
      case 0xC2:
        if (have_frame || parse_sof(d, p, plen, 1)) return 1;
        have_frame = 1;
        if (headers_only) return 0;
        break;
      case 0xC3: case 0xC5: case 0xC6: case 0xC7: case 0xC9: case 0xCA: case 0xCB: case 0xCD:
      case 0xCE: case 0xCF:
        return 2; /* lossless / arithmetic: unsupported */
      case 0xC4: if (parse_dht(d, p, plen)) return 1; break;
      case 0xDB: if (parse_dqt(d, p, plen)) return 1; break;
      case 0xDD: if (plen < 2) return 1; d->restart_interval = rd16(p); break;
      case 0xE0: if (plen >= 5 && !memcmp(p, "JFIF", 5)) d->saw_jfif = 1; break;
      case 0xE1: parse_exif(d, p, plen); break;
      case 0xEE:
        if (plen >= 12 && !memcmp(p, "Adobe", 5)) { d->saw_adobe = 1; d->adobe_transform = p[11]; }
        break;
      case 0xDA: {
        if (!have_frame || plen < 1) return 1;
        int ns = p[0];
        if (ns < 1 || ns > d->ncomp || plen < 1 + 2 * ns + 3) return 1;
        d->scan_ncomp = ns;
        for (int i = 0; i < ns; i++) {
          int cid = p[1 + 2 * i], found = -1;
          for (int j = 0; j < d->ncomp; j++) if (d->comp[j].id == cid) found = j;
          if (found < 0) return 1;
          d->scan_comp[i] = found;
          d->scan_td[i] = (p[2 + 2 * i] >> 4) & 3;
          d->scan_ta[i] = p[2 + 2 * i] & 3;
        }
        d->Ss = p[1 + 2 * ns]; d->Se = p[2 + 2 * ns];
        d->Ah = p[3 + 2 * ns] >> 4; d->Al = p[3 + 2 * ns] & 15;
        if (!d->progressive) { d->Ss = 0; d->Se = 63; d->Ah = d->Al = 0; }
        if (d->Se > 63 || d->Ss > d->Se) return 1;
        decode_scan(d);
        break;
      }
      default: break;
    }
  }
}

/* Header probe.  info: width, height, ncomp, progressive, hmax, vmax, orientation, (h,v) x4 */
int orc_jpeg_info(const uint8_t *data, size_t size, int *info) {
  dec_t *d = (dec_t *)calloc(1, sizeof(dec_t));
  d->data = data; d->size = size;
  int rc = parse_until_eoi(d, 1);
  /* orientation may follow SOF only in odd files; the common APP1-before-SOF order is covered */
  if (!rc) {
    info[0] = d->width; info[1] = d->height; info[2] = d->ncomp; info[3] = d->progressive;
    info[4] = d->hmax; info[5] = d->vmax; info[6] = d->orientation;
    for (int i = 0; i < 4; i++) { info[7 + 2 * i] = d->comp[i].h; info[8 + 2 * i] = d->comp[i].v; }
  }
  free_dec(d); free(d);
  return rc;
}

/*
 * Full decode to interleaved RGB u8 [H][W][3] (grayscale replicated, like I_RGB output).
 * Optionally returns the dequantisation-ready coefficient planes through `coef_out[c]`
 * (caller-allocated, [hblk][wblk][64] int16) and quant tables through `qt_out` (4x64 u16).
 */
int orc_jpeg_decode_rgb(const uint8_t *data, size_t size, uint8_t *rgb, int16_t **coef_out,
                        uint16_t *qt_out) {
  dec_t *d = (dec_t *)calloc(1, sizeof(dec_t));
  d->data = data; d->size = size;
  int rc = parse_until_eoi(d, 0);
  if (rc) { free_dec(d); free(d); return rc; }
  int W = d->width, H = d->height;
  for (int ci = 0; ci < d->ncomp; ci++) {
    comp_t *c = &d->comp[ci];
    if (!d->qt_present[c->tq]) { free_dec(d); free(d); return 1; }
    int stride = c->wblk * 8;
    for (int by = 0; by < c->hblk; by++)
      for (int bx = 0; bx < c->wblk; bx++)
        orc_idct_islow(c->coef + ((size_t)by * c->wblk + bx) * 64, d->qt[c->tq],
                       c->plane + (size_t)by * 8 * stride + bx * 8, stride);
    if (coef_out && coef_out[ci])
      memcpy(coef_out[ci], c->coef, sizeof(int16_t) * 64 * (size_t)c->wblk * c->hblk);
  }
  if (qt_out) memcpy(qt_out, d->qt, sizeof(d->qt));
  size_t n = (size_t)W * H;
  if (d->ncomp == 1) {
    const comp_t *c = &d->comp[0];
    for (int y = 0; y < H; y++)
      for (int x = 0; x < W; x++) {
        uint8_t v = c->plane[(size_t)y * c->wblk * 8 + x];
        uint8_t *o = rgb + ((size_t)y * W + x) * 3;
        o[0] = o[1] = o[2] = v;
      }
  } else if (d->ncomp == 3) {
    uint8_t *full = (uint8_t *)malloc(3 * n);
    for (int ci = 0; ci < 3; ci++) upsample_component(d, &d->comp[ci], full + ci * n);
    /* colour space: jdapimin.c default_decompress_parms */
    int is_rgb = 0;
    if (d->saw_jfif) is_rgb = 0;
    else if (d->saw_adobe) is_rgb = d->adobe_transform == 0;
    else is_rgb = d->comp[0].id == 'R' && d->comp[1].id == 'G' && d->comp[2].id == 'B';
    if (is_rgb) {
      for (size_t i = 0; i < n; i++) { rgb[3 * i] = full[i]; rgb[3 * i + 1] = full[n + i]; rgb[3 * i + 2] = full[2 * n + i]; }
    } else {
      ycc_to_rgb(full, full + n, full + 2 * n, rgb, n);
    }
    free(full);
  } else {
    free_dec(d); free(d);
    return 3; /* CMYK / YCCK: off the hot path */
  }
  free_dec(d); free(d);
  return 0;
}
