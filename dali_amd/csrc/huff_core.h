// Per-lane core of the GPU Huffman entropy decoder (jpeg_huffman.hip), written so that the SAME code compiles for the
// device and for the host: tools/huff_model.cpp replays the whole pipeline (tables, synchronisation with block-start
// lists, DC pass, block decode) lane by lane on the CPU and is checked against the host entropy decoder in the CPU
// test-suite, so the logic of the kernels is exercised without a GPU.  Everything that needs the machine (LDS
// staging, wave shuffles, barriers, the IDCT) stays in jpeg_huffman.hip.
#ifndef DALI_AMD_CSRC_HUFF_CORE_H_
#define DALI_AMD_CSRC_HUFF_CORE_H_
#include <cstdint>

#if defined(__HIPCC__)
#define HUFF_HD __host__ __device__ __forceinline__
#else
#define HUFF_HD inline
#endif
// A value that is the same in every lane of the wave: kept in a scalar register on the device.  Besides saving vector
// registers this matters for the waits: a value loaded from LDS right before a loop that itself writes to LDS would
// otherwise be waited for (lgkmcnt(0), i.e. for the loop's own store as well) at its use in EVERY iteration.
#if defined(__HIP_DEVICE_COMPILE__)
#define HUFF_UNIFORM(x) ((uint32_t)__builtin_amdgcn_readfirstlane((int)(x)))
#else
#define HUFF_UNIFORM(x) ((uint32_t)(x))
#endif

namespace daliamd {

#ifndef DALIAMD_FAST_BITS
#define DALIAMD_FAST_BITS 11
#endif
#ifndef DALIAMD_L2_ENTRIES
#define DALIAMD_L2_ENTRIES 512
#endif
constexpr int kFastBits = DALIAMD_FAST_BITS;    // first-level window of every code table
constexpr int kL2Entries = DALIAMD_L2_ENTRIES;  // direct second-level table for the codes longer than kFastBits
constexpr int kSyncGroup = 3;    // symbols one look-up of the position-only pass may step over

// Table entry: bits 0-6 zig-zag advance (1..64), 7-11 bits consumed (code length + magnitude bits s), 12-15 s.
//   DC symbol (category s):  advance 1
//   AC symbol (run r, size s): s != 0: r + 1;  ZRL (0xF0): 16;  any other s == 0 (EOB): 64 = "to the end of the block"
HUFF_HD uint32_t MakeEntry(int len, int sym, bool is_dc) {
  int s = sym & 15, r = sym >> 4;
  int adv = is_dc ? 1 : (s ? r + 1 : (r == 15 ? 16 : 64));
  return (uint32_t)((s << 12) | ((len + s) << 7) | adv);
}
HUFF_HD uint32_t SyncGroup(uint32_t z, uint32_t used, uint32_t count) { return z | (used << 7) | (count << 12); }

// Code tables of the value-extracting passes (DC pass, block decode).
struct HuffTables {
  uint16_t fast[4][1 << kFastBits];  // [0],[1] = DC tables 0,1; [2],[3] = AC tables 0,1
  uint16_t l2[4][kL2Entries];        // codes longer than kFastBits, indexed by (16-bit code window) - l2_first
  int32_t l2_first[4];
  int32_t l2_size[4];                // entries in use; -1: the long codes span more than kL2Entries -> search
  int32_t maxcode[4][18];            // canonical-code search tables (T.81 F.2.2.3), the fallback
  int32_t valoff[4][18];
  uint8_t vals[4][256];
  uint32_t dc_mask, ac_mask;         // bit k: table selector of the k-th block of the MCU
  int32_t bpm, reserved;
};
static_assert(sizeof(HuffTables) % 16 == 0, "copied with 16-byte accesses");

// Tables of the position-only passes (SyncKernel / PropagateKernel).  They do not extract values, so one look-up may
// step over a GROUP of up to three symbols of one block: a 32-bit entry holds, for the kFastBits-bit window,
//   bits  0-13  the whole group:  z advance (7 bits) | bits used << 7 (5 bits) | symbol count << 12 (1..3)
//   bits 14-23  what precedes the group's LAST symbol: z advance (6 bits) | bits used << 6 (4 bits); zero for a
//               single symbol.  The group may be taken when these symbols leave the block open - decided per step
//   bits 24-31  groups of three only: the first symbol alone, bits used (4 bits) | (z advance - 1) << 4, for the
//               (rare) step that cannot take the group; with two symbols the fields above already describe it
// A group continues behind a symbol when that one is not an end-of-block and the CODE of the next lies inside the
// window behind it (its magnitude bits need not).  DC entries continue into the AC table of their block when all the
// blocks that use the DC table use the same AC table.
struct SyncTables {
  uint32_t t32[4][1 << kFastBits];   // [0],[1] = DC tables 0,1; [2],[3] = AC tables 0,1; 0 = code longer than the window
  uint16_t l2[4][kL2Entries];
  int32_t l2_first[4];
  int32_t l2_size[4];
  int32_t maxcode[4][18];
  int32_t valoff[4][18];
  uint8_t vals[4][256];
  uint32_t dc_mask, ac_mask;
  int32_t bpm, reserved;
};
static_assert(sizeof(SyncTables) % 16 == 0, "copied with 16-byte accesses");

// ------------------------------------------------------------------------------------------------ table construction
// Canonical code assignment (ITU-T T.81 Annex C) of one table: per code length the largest code and the symbol
// offset; range of the 16-bit windows of the codes longer than kFastBits.
HUFF_HD void CodeRanges(const uint8_t bits[16], int32_t maxcode[18], int32_t valoff[18], int32_t *l2_first_out,
                        int32_t *l2_size_out) {
  int code = 0, p = 0;
  int l2_first = 1 << 16, l2_end = 0;
  maxcode[0] = -1; valoff[0] = 0; maxcode[17] = -1; valoff[17] = 0;
  for (int l = 1; l <= 16; l++) {
    const int n = bits[l - 1];
    valoff[l] = p - code;
    if (n && l > kFastBits) {
      if (l2_end == 0) l2_first = (code << (16 - l)) & 0xFFFF;
      l2_end = ((code + n) << (16 - l));  // one past the last 16-bit window of the codes seen so far
    }
    p += n;
    code += n;
    maxcode[l] = n ? code - 1 : -1;
    code <<= 1;
  }
  const int size = l2_end ? l2_end - l2_first : 0;
  *l2_first_out = l2_first;
  *l2_size_out = size <= kL2Entries ? size : -1;  // -1: too spread out for the direct table, LongCode searches
}

// First-level entry for the kFastBits-bit window w: the shortest length whose code range contains the window's prefix
// (0: the code is longer than the window).
template <typename Tables>
HUFF_HD uint16_t FastEntry(const Tables &L, int t, int w) {
  for (int l = 1; l <= kFastBits; l++) {
    const int cd = w >> (kFastBits - l);
    if (cd <= L.maxcode[t][l]) return (uint16_t)MakeEntry(l, L.vals[t][(cd + L.valoff[t][l]) & 255], t < 2);
  }
  return 0;
}
// Second-level entry j (16-bit window l2_first + j).
template <typename Tables>
HUFF_HD uint16_t L2Entry(const Tables &L, int t, int j) {
  if (j >= L.l2_size[t]) return 0;
  const int w = L.l2_first[t] + j;
  for (int l = kFastBits + 1; l <= 16; l++) {
    const int cd = w >> (16 - l);
    if (cd <= L.maxcode[t][l]) return (uint16_t)MakeEntry(l, L.vals[t][(cd + L.valoff[t][l]) & 255], t < 2);
  }
  return 0;
}
// Symbol-group entry of the position-only tables for window w of table tb (needs the finished fast[] tables).
HUFF_HD uint32_t SyncEntry(const HuffTables &L, int tb, int w) {
  const uint32_t e1 = L.fast[tb][w];
  if (!e1) return 0;
  // table the block continues with: an AC table itself, or the one AC table every block of this DC table uses
  int ac = tb;
  if (tb < 2) {
    ac = -1;
    for (int k = 0; k < L.bpm; k++) {
      if ((int)((L.dc_mask >> k) & 1u) != tb) continue;
      const int a = 2 + (int)((L.ac_mask >> k) & 1u);
      ac = ac == -1 || ac == a ? a : -2;
    }
  }
  uint32_t z = e1 & 127, used = (e1 >> 7) & 31, count = 1, zprev = 0, uprev = 0;
  const uint32_t z1 = z, u1 = used;
  while (ac >= 2 && count < (uint32_t)kSyncGroup && z < 64 && used < (uint32_t)kFastBits) {
    const uint32_t e2 = L.fast[ac][(w << used) & ((1 << kFastBits) - 1)];
    const uint32_t z2 = e2 & 127, u2 = (e2 >> 7) & 31, len2 = u2 - (e2 >> 12);
    if (!e2 || used + len2 > (uint32_t)kFastBits) break;  // the next code is not determined by the window
    zprev = z; uprev = used;
    z += z2; used += u2; count++;
  }
  uint32_t e = SyncGroup(z, used, count) | (zprev << 14) | (uprev << 20);
  if (count == 3) e |= (u1 << 24) | ((z1 - 1) << 28);
  return e;
}

// ------------------------------------------------------------------------------------------------ decoder state
struct DecodeState {
  uint32_t pos;  // bit position of the next symbol
  uint32_t c;    // block index inside the MCU
  uint32_t z;    // zig-zag index of the next coefficient (0 = DC)
};
HUFF_HD uint64_t Pack(const DecodeState &s) { return ((uint64_t)s.pos << 16) | ((uint64_t)s.c << 8) | (uint64_t)s.z; }
HUFF_HD DecodeState Unpack(uint64_t v) {
  return DecodeState{(uint32_t)(v >> 16), (uint32_t)((v >> 8) & 255), (uint32_t)(v & 255)};
}
constexpr uint64_t kNoState = ~0ull;  // unpacks to a position past any stream

// Restart intervals (T.81 B.2.4.4 DRI, E.2.4, F.1.2.3): every `interval` MCUs the encoder pads the stream to a byte
// boundary with one-bits, writes an RSTn marker and resets the DC predictions.  The un-stuffing pass removes the markers
// and notes the CLEAN byte offset behind each one (`pos`, ascending, `n` of them): there the next interval starts with
// the decoder state (c = 0, z = 0) whatever came before - a synchronisation point that needs no relaxation.
template <typename RstPos>
struct RestartView {
  RstPos pos;
  int n;
  uint32_t interval;  // MCUs per interval; 0: the stream has none
  int hint;           // index of the first boundary that can matter to the caller (at or behind its slice)
};

HUFF_HD uint32_t Bswap32(uint32_t v) { return __builtin_bswap32(v); }

// Rare path: the code is longer than kFastBits bits (or is not a code at all).  Out of line on the device.
template <typename Tables>
#if defined(__HIPCC__)
__host__ __device__ __noinline__
#else
inline
#endif
uint32_t LongCode(const Tables &L, uint32_t slot, uint32_t peek, bool is_dc) {
  const uint32_t code16 = peek >> 16;
  uint32_t e = 0;
  const int size = L.l2_size[slot];
  if (size >= 0) {
    const int idx = (int)code16 - L.l2_first[slot];
    if (idx >= 0 && idx < size) e = L.l2[slot][idx];
  } else {
    for (int l = kFastBits + 1; l <= 16; l++) {
      int cd = (int)(code16 >> (16 - l));
      if (cd <= L.maxcode[slot][l]) {
        e = MakeEntry(l, L.vals[slot][(cd + L.valoff[slot][l]) & 255], is_dc);
        break;
      }
    }
  }
  // not a code (garbage start state, or the padding behind the stream): consume 16 bits, decode nothing
  return e ? e : MakeEntry(16, 0, is_dc);
}

// ------------------------------------------------------------------------------------------------ position-only pass
// Decodes the groups of symbols that start in [st.pos, end_bits) - positions only - and reports where every block
// STARTS through `store(j, rem)`: the j-th block found starts `rem` bits before end_bits (negative: a group may run a
// few symbols past the end).  A block starts where the previous one ends; the very first block of a stream starts
// at bit 0, which only the decode that begins there sees.  Returns the number of block starts.
//
// Every step reports the position it reached for the CURRENT index - store(j, rem, ended) - and the index only
// advances when the step ended a block (`ended`; the last report of an index is the one that counts).  A store into
// LDS simply writes every time: no branch, no execution mask, off the dependency chain of the step.  A store that is
// expensive (global memory) looks at `ended`.
// Stream dword at BYTE offset kb (a multiple of 4).  On the device the offset stays a 32-bit register next to a
// uniform base (no 64-bit address arithmetic in the decode loop).
template <typename Words>
HUFF_HD uint32_t WordAtByte(Words words, uint32_t kb) { return words[kb >> 2]; }
#if defined(__HIPCC__)
__device__ __forceinline__ uint32_t WordAtByte(const uint32_t __attribute__((address_space(1))) *words, uint32_t kb) {
  return *(const uint32_t __attribute__((address_space(1))) *)((const uint8_t __attribute__((address_space(1))) *)words + kb);
}
#endif

//
// RST: the stream has restart intervals (`rst`).  (1) A decode that has just finished an MCU less than 8 bits in front
// of a boundary, with nothing but one-bits in between, has finished the interval: a complete symbol always holds a
// zero bit (no code word is all ones), so those bits cannot be another MCU - they are the padding.  It steps over them
// and goes on at the boundary.  (2) A decode that finds itself BEHIND a boundary it did not arrive at that way either
// started from a wrong guess (it continues from the boundary with the state every interval starts in: each boundary is
// a point where guessed states become true ones) or the stream's padding is not the one-bits of the standard;
// `*crossed` reports it, and the caller refuses the stream when that happens to a decode that started from the truth.
template <bool RST, typename Tables, typename Words, typename RstPos, typename Store>
HUFF_HD int SyncDecodeRangeT(const Tables &L, Words words, DecodeState &st, uint32_t end_bits, const RestartView<RstPos> &rst,
                             Store store, bool *crossed) {
  int nb = 0;
  uint32_t c = st.c, z = st.z;
  int rem = (int)(end_bits - st.pos);  // bits left before the end of the slice (<= 0: done)
  constexpr int kFar = 1 << 30;
  int ri = 0, rem_r = kFar;            // next boundary: its index, the bits up to it
  bool dirty = false;
  if (RST) {
    ri = rst.hint;
    while (ri < rst.n && rst.pos[ri] <= (st.pos >> 3)) ri++;   // first boundary behind the position (hint: 0-1 steps)
    if (ri < rst.n) rem_r = (int)(rst.pos[ri] * 8u - st.pos);
  }
  if (st.pos == 0) {                   // the stream's first block
    store(0, rem, true);
    nb = 1;
  }
  // bit window: hi:lo = stream bits [32k, 32k+64), `off` of hi's bits already consumed; the following dword is in flight
  uint32_t kb = (st.pos >> 5) << 2;  // byte offset of hi's dword
  uint32_t off = st.pos & 31;
  uint32_t hi = Bswap32(WordAtByte(words, kb)), lo = Bswap32(WordAtByte(words, kb + 4)), nxt = WordAtByte(words, kb + 8);
  const uint32_t dc_mask = HUFF_UNIFORM(L.dc_mask), ac_mask = HUFF_UNIFORM(L.ac_mask), bpm = HUFF_UNIFORM(L.bpm);
  // The report of a step is handed to `store` right BEHIND the table look-up of the next step: an LDS store issued
  // in front of the look-up would sit in front of it in the (in-order) LDS queue and add its service time to the
  // dependency chain of every step; behind it, it drains while the step computes.
  int s_nb = nb, s_rem = rem;
  bool s_ended = false;
  while (rem > 0) {
    const uint32_t peek = (uint32_t)(((((uint64_t)hi << 32) | lo) << off) >> 32);
    const bool is_dc = z == 0;
    const uint32_t slot = (((is_dc ? dc_mask : ac_mask) >> c) & 1u) + (is_dc ? 0u : 2u);
    uint32_t e = L.t32[slot][peek >> (32 - kFastBits)];
    store(s_nb, s_rem, s_ended);
    if (__builtin_expect(e == 0, 0)) {
      const uint32_t e16 = LongCode(L, slot, peek, is_dc);
      e = SyncGroup(e16 & 127, (e16 >> 7) & 31, 1);
    }
    // The group may be taken when the symbols before its last one leave the block open.  (Round 1 also required the
    // last symbol to start inside the range; measured, that costs more per step than it saves in relaxation rounds -
    // the rounds of the bench set are identical with and without it.)
    const int zprev = (int)((e >> 14) & 63), uprev = (int)((e >> 20) & 15);
    const int ok = (int)z + zprev - 64;
    uint32_t used = (e >> 7) & 31, zinc = e & 127;
    if (__builtin_expect(ok >= 0, 0)) {  // rare: take the first symbol only
      const bool three = ((e >> 12) & 3) == 3;
      used = three ? (e >> 24) & 15 : (uint32_t)uprev;
      zinc = three ? ((e >> 28) & 15) + 1 : (uint32_t)zprev;
    }
    rem -= (int)used;
    off += used;
    z += zinc;
    if (off >= 32) {
      hi = lo;
      lo = Bswap32(nxt);
      nxt = WordAtByte(words + 3, kb);  // dword k + 3 of the window that starts at k (uniform base, 32-bit offset as it is)
      kb += 4;
      off -= 32;
    }
    const bool end_of_block = z >= 64;
    s_nb = nb;
    s_rem = rem;
    s_ended = end_of_block;
    const uint32_t c1 = c + 1 == bpm ? 0 : c + 1;
    z = end_of_block ? 0 : z;
    c = end_of_block ? c1 : c;
    nb += end_of_block ? 1 : 0;
    if (RST) {
      rem_r -= (int)used;
      if (__builtin_expect(rem_r < 8, 0)) {   // (rem_r stays far away behind the last boundary)
        bool jump = rem_r < 0;
        dirty = dirty || jump;
        if (!jump && end_of_block && c1 == 0) {   // an MCU ends here: only padding up to the boundary?
          const uint32_t ahead = (uint32_t)(((((uint64_t)hi << 32) | lo) << off) >> 32);
          const uint32_t zeros = rem_r ? ~ahead >> (32 - rem_r) : 0u;   // zero bits among the rem_r bits ahead
          jump = zeros == 0;
        }
        if (jump) {
          // on to the boundary: the block that ended here (or that the boundary cuts off) is followed by one that
          // starts AT the boundary
          if (!end_of_block) nb++;
          rem -= rem_r;
          c = 0; z = 0;
          s_nb = nb - 1; s_rem = rem; s_ended = true;
          const uint32_t at = end_bits - (uint32_t)rem;
          kb = (at >> 5) << 2;
          off = at & 31;
          hi = Bswap32(WordAtByte(words, kb)); lo = Bswap32(WordAtByte(words, kb + 4)); nxt = WordAtByte(words, kb + 8);
          ri++;
          rem_r = ri < rst.n ? (int)(rst.pos[ri] * 8u - at) : kFar;
        }
      }
    }
  }
  store(s_nb, s_rem, s_ended);
  st.pos = end_bits - (uint32_t)rem;
  st.c = c;
  st.z = z;
  if (RST) *crossed = dirty;
  return nb;
}
template <typename Tables, typename Words, typename Store>
HUFF_HD int SyncDecodeRange(const Tables &L, Words words, DecodeState &st, uint32_t end_bits, Store store) {
  return SyncDecodeRangeT<false>(L, words, st, end_bits, RestartView<const uint32_t *>{nullptr, 0, 0, 0}, store, nullptr);
}
// the stream's own variant (rst.interval decides; uniform per stream)
template <typename Tables, typename Words, typename RstPos, typename Store>
HUFF_HD int SyncDecodeRange(const Tables &L, Words words, DecodeState &st, uint32_t end_bits, const RestartView<RstPos> &rst,
                            Store store, bool *crossed) {
  *crossed = false;
  if (rst.interval) return SyncDecodeRangeT<true>(L, words, st, end_bits, rst, store, crossed);
  return SyncDecodeRangeT<false>(L, words, st, end_bits, rst, store, crossed);
}

// ------------------------------------------------------------------------------------------------ value passes
// (declared here for the indexed pass below; defined with the value passes)
HUFF_HD int Extend(uint32_t bits, uint32_t s);

// ------------------------------------------------------------------------------------------------ indexed pass
// Side information of a RESIDENT stream (round 5): what the relaxation of the position pass finds out about a 256-byte
// slice of the clean stream does not change from epoch to epoch, so the first decode of a stream that stays in HBM
// keeps it - 12 bytes per slice, 4.7 % of the stream - and every later decode starts each slice from the truth:
//   w0  bits 0-25  ordinal of the first block whose DC symbol lies in the slice's decode | bits 26-31 zig-zag index at entry
//   w1  bits 0-11  entry position, in bits behind the slice's first bit | 12-15 block index inside the MCU | 16-31 DC level
//                  of component 0 in front of that block (mod 2^16: only differences and 16-bit results are used)
//   w2  DC levels of components 1 and 2 the same way
// One decode per slice, no relaxation, no hand-over check, no separate DC pass (the predictors are known), and - the
// ordinals being known - a slice whose blocks a region-of-interest decode does not need is not decoded at all.
struct SliceIndex {
  uint32_t w0, w1, w2;
};
constexpr uint32_t kIndexOrdinalMask = (1u << 26) - 1u;
HUFF_HD SliceIndex PackSliceIndex(uint32_t first_block, const DecodeState &rel, const uint32_t dc[3]) {
  return SliceIndex{(first_block & kIndexOrdinalMask) | (rel.z << 26), (rel.pos & 4095u) | ((rel.c & 15u) << 12) | (dc[0] << 16),
                    (dc[1] & 0xFFFFu) | (dc[2] << 16)};
}
HUFF_HD uint32_t IndexFirstBlock(const SliceIndex &e) { return e.w0 & kIndexOrdinalMask; }
// entry state with the position relative to the slice's first bit
HUFF_HD DecodeState IndexEntryState(const SliceIndex &e) { return DecodeState{e.w1 & 4095u, (e.w1 >> 12) & 15u, e.w0 >> 26}; }

// Decodes the slice [st.pos, end_bits) once from its TRUE entry state: positions with the symbol-group tables `L` as the
// relaxation does, and at every block start the DC symbol with the value tables `D` (one more look-up in the same
// step, off the chain).  emit(c, pos_behind_dc, diff): block index inside the MCU, bit position of the block's first AC
// symbol, DC difference - in stream order, for every block whose DC symbol starts in front of end_bits.
template <typename STables, typename DTables, typename Words, typename Emit>
HUFF_HD void IndexedDecodeSlice(const STables &L, const DTables &D, Words words, const DecodeState &st, uint32_t end_bits,
                                Emit emit) {
  uint32_t c = st.c, z = st.z;
  int rem = (int)(end_bits - st.pos);
  uint32_t kb = (st.pos >> 5) << 2;  // byte offset of hi's dword
  uint32_t off = st.pos & 31;
  uint32_t hi = Bswap32(WordAtByte(words, kb)), lo = Bswap32(WordAtByte(words, kb + 4)), nxt = WordAtByte(words, kb + 8);
  const uint32_t dc_mask = HUFF_UNIFORM(L.dc_mask), ac_mask = HUFF_UNIFORM(L.ac_mask), bpm = HUFF_UNIFORM(L.bpm);
  while (rem > 0) {
    const uint32_t peek = (uint32_t)(((((uint64_t)hi << 32) | lo) << off) >> 32);
    const bool is_dc = z == 0;
    const uint32_t sel = (dc_mask >> c) & 1u;
    const uint32_t slot = is_dc ? sel : ((ac_mask >> c) & 1u) + 2u;
    uint32_t e = L.t32[slot][peek >> (32 - kFastBits)];
    if (is_dc) {
      uint32_t ed = D.fast[sel][peek >> (32 - kFastBits)];
      if (__builtin_expect(ed == 0, 0)) ed = LongCode(D, sel, peek, true);
      const uint32_t used1 = (ed >> 7) & 31, s = ed >> 12;
      emit(c, end_bits - (uint32_t)rem + used1, Extend(peek >> (32 - used1), s));
    }
    if (__builtin_expect(e == 0, 0)) {
      const uint32_t e16 = LongCode(L, slot, peek, is_dc);
      e = SyncGroup(e16 & 127, (e16 >> 7) & 31, 1);
    }
    const int zprev = (int)((e >> 14) & 63), uprev = (int)((e >> 20) & 15);
    const int ok = (int)z + zprev - 64;
    uint32_t used = (e >> 7) & 31, zinc = e & 127;
    if (__builtin_expect(ok >= 0, 0)) {  // rare: take the first symbol only (SyncDecodeRangeT)
      const bool three = ((e >> 12) & 3) == 3;
      used = three ? (e >> 24) & 15 : (uint32_t)uprev;
      zinc = three ? ((e >> 28) & 15) + 1 : (uint32_t)zprev;
    }
    rem -= (int)used;
    off += used;
    z += zinc;
    if (off >= 32) {
      hi = lo;
      lo = Bswap32(nxt);
      nxt = WordAtByte(words + 3, kb);
      kb += 4;
      off -= 32;
    }
    const bool end_of_block = z >= 64;
    const uint32_t c1 = c + 1 == bpm ? 0 : c + 1;
    z = end_of_block ? 0 : z;
    c = end_of_block ? c1 : c;
  }
}

// T.81 F.2.2.1 EXTEND: the low s bits of `bits` are the magnitude bits m -> value; s == 0 gives 0.
HUFF_HD int Extend(uint32_t bits, uint32_t s) {
  const uint32_t full = (1u << s) - 1u;  // 2^s - 1
  const uint32_t m = bits & full;
  return (int)m - (int)(m <= (full >> 1) ? full : 0u);
}
// 32 stream bits starting at bit `pos` (big-endian bit order; `words` = the clean stream as little-endian dwords)
template <typename Words>
HUFF_HD uint32_t Peek32(Words words, uint32_t pos) {
  const uint32_t k = pos >> 5, off = pos & 31;
  const uint32_t hi = Bswap32(words[k]), lo = Bswap32(words[k + 1]);
  return (uint32_t)(((((uint64_t)hi << 32) | lo) << off) >> 32);
}

// DC symbol of the block that starts at bit `pos`, decoded with DC table `sel`: returns the difference, *used = bits
// consumed (code + magnitude: at most 16 + 11 bits, so one 32-bit window is enough).
template <typename Tables, typename Words>
HUFF_HD int DecodeDc(const Tables &L, Words words, uint32_t pos, uint32_t sel, uint32_t *used_out) {
  const uint32_t peek = Peek32(words, pos);
  uint32_t e = L.fast[sel][peek >> (32 - kFastBits)];
  if (__builtin_expect(e == 0, 0)) e = LongCode(L, sel, peek, true);
  const uint32_t used = (e >> 7) & 31, s = e >> 12;
  *used_out = used;
  return Extend(peek >> (32 - used), s);
}

// Bit window over the clean stream: hi:lo = stream bits [32k, 32k+64), `off` of hi's bits already consumed; the dword
// behind them is in flight (`nxt`, still little-endian).  Opening a window only issues its three loads, so a caller
// can open the window of its NEXT piece of work before it finishes the current one.
struct BitWindow {
  int k;
  uint32_t off, hi, lo, nxt;
};
template <typename Words>
HUFF_HD BitWindow OpenWindow(Words words, uint32_t pos) {
  BitWindow w;
  w.k = (int)(pos >> 5);
  w.off = pos & 31;
  w.hi = words[w.k];  // byte-swapped at first use
  w.lo = words[w.k + 1];
  w.nxt = words[w.k + 2];
  return w;
}

// AC coefficients of one block: symbols from the window's position (just behind the DC symbol) until the block is full
// or an end-of-block arrives.  coef[z] (z = zig-zag index 1..63) receives the values; coef[64] is a scratch slot that
// swallows what does not carry a coefficient (end-of-block, ZRL past the end, a run that overshoots the block in a
// corrupt stream), so the store needs no condition.  The block must be zero-filled by the caller.
template <typename Tables, typename Words, typename Coef>
HUFF_HD void DecodeBlockAc(const Tables &L, Words words, const BitWindow &win, uint32_t ac_slot, Coef coef) {
  uint32_t kb = (uint32_t)win.k << 2;  // byte offset of hi's dword
  uint32_t off = win.off;
  uint32_t hi = Bswap32(win.hi), lo = Bswap32(win.lo), nxt = win.nxt;
  uint32_t z = 1;
  while (z < 64) {
    const uint32_t peek = (uint32_t)(((((uint64_t)hi << 32) | lo) << off) >> 32);
    uint32_t e = L.fast[ac_slot][peek >> (32 - kFastBits)];
    if (__builtin_expect(e == 0, 0)) e = LongCode(L, ac_slot, peek, false);
    const uint32_t used = (e >> 7) & 31, s = e >> 12, adv = e & 127;
    const int val = Extend(peek >> (32 - used), s);
    z += adv;                                   // index behind the coefficient this symbol carries
    const uint32_t at = z - 1 < 64 ? z - 1 : 64;  // 64: nothing to store (s == 0 stores a zero where it lands: harmless,
    coef[at] = (int16_t)val;                    // the positions of a block are visited in increasing order)
    off += used;
    if (off >= 32) {
      hi = lo;
      lo = Bswap32(nxt);
      nxt = WordAtByte(words + 3, kb);
      kb += 4;
      off -= 32;
    }
  }
}
template <typename Tables, typename Words, typename Coef>
HUFF_HD void DecodeBlockAc(const Tables &L, Words words, uint32_t pos, uint32_t ac_slot, Coef coef) {
  DecodeBlockAc(L, words, OpenWindow(words, pos), ac_slot, coef);
}

// zig-zag scan order expressed in column-major block positions (= the transposed zig-zag): coefficient z of the scan
// sits at column kZigZagColMajorHost[z] / 8, row kZigZagColMajorHost[z] % 8
constexpr uint8_t kZigZagColMajorTable[64] = {
    0, 8, 1, 2, 9, 16, 24, 17, 10, 3, 4, 11, 18, 25, 32, 40, 33, 26, 19, 12, 5, 6, 13, 20, 27, 34, 41, 48, 56, 49, 42, 35,
    28, 21, 14, 7, 15, 22, 29, 36, 43, 50, 57, 58, 51, 44, 37, 30, 23, 31, 38, 45, 52, 59, 60, 53, 46, 39, 47, 54, 61, 62,
    55, 63};

}  // namespace daliamd
#endif  // DALI_AMD_CSRC_HUFF_CORE_H_
