// Host model of the GPU entropy decoder (dali_amd/csrc/jpeg_huffman.hip).  It runs the SAME per-lane code
// (dali_amd/csrc/huff_core.h: table construction, position-only decode with block-start lists, DC decode, block
// decode) and restates the orchestration of the kernels lane by lane - relaxation rounds with capped LDS lists and
// the overflow path, the dense per-segment start lists, segment hand-over with repair, block ordinals, the DC pass
// with per-component prefix sums and segment totals, the task / class mapping of the block pass - on the CPU, then
// compares every coefficient with the host entropy decoder.  Built and run by tests/test_huff_model.py (no GPU).
//
//   g++ -O2 -std=c++17 -shared -fPIC -Iinclude -Idali_amd/csrc tools/huff_model.cpp -Ldali_amd/lib -ldali_amd_host
//     -Wl,-rpath,$PWD/dali_amd/lib -o /tmp/libhuff_model.so
#include <algorithm>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "dali_amd_host.h"
#include "huff_core.h"

using namespace daliamd;

namespace {

// the constants of jpeg_huffman.hip (the model takes them as parameters so that tests can shrink them and force the
// rare paths: list overflow, failed warm-up, many segments)
struct Params {
  int slice_bytes = 256, seg_threads = 256, warm_lanes = 12, list_cap = 65, blocks_per_wg = 384;
};

struct Lane {
  uint32_t begin = 0, end = 0;
  bool active = false;
  uint64_t in = kNoState, out = kNoState;
  int nstart = 0;
  bool crossed = false;
  std::vector<uint16_t> list;  // cap + 1 slots
};

struct SegRec {
  uint64_t out = kNoState;
  int nstart_total = 0, block_base = 0;
  int dc_total[3] = {0, 0, 0};
  bool crossed = false;
  std::vector<uint32_t> starts;
  std::vector<Lane> lanes;  // the segment's own lanes (LaneRec)
};

Lane MakeLane(long long slice_index, uint32_t total_bits, const Params &P) {
  Lane ln;
  ln.list.assign(P.list_cap + 1, 0);
  if (slice_index < 0) return ln;
  const unsigned long long b = (unsigned long long)slice_index * (P.slice_bytes * 8ull);
  ln.begin = (uint32_t)std::min<unsigned long long>(b, total_bits);
  ln.end = (uint32_t)std::min<unsigned long long>(b + P.slice_bytes * 8ull, total_bits);
  ln.active = ln.begin < total_bits;
  return ln;
}

struct Model {
  Params P;
  std::vector<uint32_t> words;  // clean stream as little-endian dwords
  uint32_t total_bits = 0;
  HuffTables H;
  SyncTables S;
  std::vector<uint32_t> rst_pos;  // clean byte offsets at which the restart intervals start (UnstuffScatterKernel)
  uint32_t restart_interval = 0;
  int rounds_max = 0, repairs = 0, overflow_lanes = 0, jumps = 0;

  // MakeRstView: hint = the first boundary at or behind the lane's slice
  RestartView<const uint32_t *> Rst(uint32_t begin_bits) const {
    RestartView<const uint32_t *> v{rst_pos.data(), (int)rst_pos.size(), restart_interval, 0};
    if (restart_interval)
      v.hint = (int)(std::lower_bound(rst_pos.begin(), rst_pos.end(), begin_bits >> 3) - rst_pos.begin());
    return v;
  }

  void Relax(std::vector<Lane> &ln, std::vector<uint64_t> &state) {
    const int T = (int)ln.size();
    for (int round = 0; round <= T; round++) {
      for (int t = 0; t < T; t++) {
        Lane &l = ln[t];
        const uint64_t ni = state[t];
        if (l.active && ni != l.in) {
          l.in = ni;
          DecodeState st = Unpack(ni);
          l.nstart = 0;
          if (st.pos < l.end) {
            const int cap = P.list_cap;
            l.nstart = SyncDecodeRange(S, words.data(), st, l.end, Rst(l.begin),
                                       [&](int nb, int rem, bool) { l.list[nb < cap ? nb : cap] = (uint16_t)rem; }, &l.crossed);
          }
          l.out = Pack(st);
        }
      }
      bool changed = false;
      for (int t = 0; t + 1 < T; t++)
        if (ln[t].active && state[t + 1] != ln[t].out) { state[t + 1] = ln[t].out; changed = true; }
      rounds_max = std::max(rounds_max, round + 1);
      if (!changed) break;
    }
  }

  // WriteSegmentStarts: lanes [first, ..) belong to the segment
  // (the lanes run concurrently on the GPU: the model takes them in DESCENDING order so that a lane writing into the
  // region of the next one - which an ascending loop would silently repair - shows up as a mismatch)
  int WriteStarts(std::vector<Lane> &ln, int first, int count, std::vector<uint32_t> &starts, int seg_cap) {
    starts.assign(seg_cap, 0xFFFFFFFFu);
    std::vector<int> bases(count + 1, 0);
    for (int t = 0; t < count; t++) bases[t + 1] = bases[t] + ln[first + t].nstart;
    for (int t = first + count - 1; t >= first; t--) {
      Lane &l = ln[t];
      const int base = bases[t - first];
      if (l.nstart <= P.list_cap) {
        for (int j = 0; j < l.nstart; j++)
          if (base + j < seg_cap) starts[base + j] = l.end - (uint32_t)(int32_t)(int16_t)l.list[j];
      } else {
        overflow_lanes++;
        DecodeState st = Unpack(l.in);
        const uint32_t end = l.end;
        const int b = base, cnt = l.nstart;
        bool crossed;
        SyncDecodeRange(S, words.data(), st, end, Rst(l.begin), [&](int nb, int rem, bool ended) {
          if (ended && nb < cnt && b + nb < seg_cap) starts[b + nb] = end - (uint32_t)rem;
        }, &crossed);
      }
    }
    return bases[count];
  }
};

std::string g_msg;
int Fail(const char *fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_msg = buf;
  return 1;
}

}  // namespace

extern "C" const char *huff_model_message() { return g_msg.c_str(); }

// Decodes `jpeg` with the model and compares with the host entropy decoder.  params: {slice_bytes, seg_threads,
// warm_lanes, list_cap, blocks_per_wg} or NULL for the kernel's constants.  stats (optional, 7 ints): segments,
// relaxation rounds (max), repaired segments, lanes that took the list-overflow path, blocks, block starts found, restart boundaries.
// Returns 0 when every coefficient matches, 1 on a mismatch / error (huff_model_message()), 2 when the stream is not
// eligible for the GPU decoder.
extern "C" int huff_model_check(const uint8_t *jpeg, size_t size, const int *params, int *stats) {
  Model M;
  if (params) {
    M.P.slice_bytes = params[0]; M.P.seg_threads = params[1]; M.P.warm_lanes = params[2]; M.P.list_cap = params[3];
    M.P.blocks_per_wg = params[4];
  }
  const Params &P = M.P;
  daliamdJpegInfo info;
  daliamdJpegScan sc;
  if (daliamdJpegParse(jpeg, size, &info) != 0) return Fail("parse: %s", daliamdHostGetLastErrorMessage());
  {
    // the operator's analysis: headers only, the segment is "the rest of the file"; the exact walk must agree on
    // everything but the length
    daliamdJpegInfo info2;
    daliamdJpegScan exact;
    if (daliamdJpegAnalyzeHeader(jpeg, size, &info2, &sc) != 0 || !sc.eligible) return 2;
    if (memcmp(&info, &info2, sizeof info)) return Fail("daliamdJpegAnalyzeHeader and daliamdJpegParse disagree");
    if (daliamdJpegAnalyzeScan(jpeg, size, &info, &exact) != 0 || !exact.eligible) return 2;
    if (!sc.length_is_upper_bound || exact.length_is_upper_bound || exact.ecs_length > sc.ecs_length)
      return Fail("scan lengths: header %lld, walk %lld", (long long)sc.ecs_length, (long long)exact.ecs_length);
    exact.ecs_length = sc.ecs_length; exact.length_is_upper_bound = 1;
    if (memcmp(&exact, &sc, sizeof sc)) return Fail("daliamdJpegAnalyzeHeader and daliamdJpegAnalyzeScan disagree");
  }
  // ---- reference: the host entropy decoder (column-major blocks) ----
  std::vector<std::vector<int16_t>> ref(info.num_components);
  int16_t *ptrs[4] = {nullptr, nullptr, nullptr, nullptr};
  for (int c = 0; c < info.num_components; c++) {
    ref[c].assign((size_t)info.coef_elems[c], 0);
    ptrs[c] = ref[c].data();
  }
  uint16_t quant[4 * 64];
  if (daliamdJpegDecodeCoefficients(jpeg, size, &info, ptrs, quant) != 0)
    return Fail("host decode: %s", daliamdHostGetLastErrorMessage());
  // ---- un-stuffing (PrepareKernel / UnstuffScatterKernel: LoadChunk's rules byte by byte): clean stream + all-ones
  // padding, the restart boundaries, the end of the segment ----
  std::vector<uint8_t> clean;
  const uint8_t *p = jpeg + sc.ecs_offset;
  bool rst_without_dri = false;
  for (int64_t i = 0; i < sc.ecs_length; i++) {
    const bool next_valid = i + 1 < sc.ecs_length;
    const uint8_t b = p[i], nb = next_valid ? p[i + 1] : 0;
    const bool after_ff = i > 0 && p[i - 1] == 0xFF;
    const bool marker = b == 0xFF && next_valid && nb != 0;
    const bool is_rst = marker && (nb & 0xF8) == 0xD0;
    if (marker && !is_rst && nb != 0xFF) break;   // the segment ends here
    if (is_rst) {
      if (sc.restart_interval) M.rst_pos.push_back((uint32_t)clean.size()); else rst_without_dri = true;
    }
    if (marker || (after_ff && (b == 0 || (b & 0xF8) == 0xD0))) continue;
    clean.push_back(b);
  }
  if (rst_without_dri) return Fail("status 3: RSTn markers without DRI");
  M.restart_interval = (uint32_t)sc.restart_interval;
  {
    const int mcus = sc.mcus_x * sc.mcus_y;
    const size_t cap = sc.restart_interval ? (size_t)(mcus + sc.restart_interval - 1) / sc.restart_interval : 0;
    if (M.rst_pos.size() > cap) M.rst_pos.resize(cap);
  }
  const uint32_t clean_len = (uint32_t)clean.size();
  M.total_bits = clean_len * 8;
  clean.resize(clean_len + 40, 0xFF);
  clean.resize((clean.size() + 512 + 3) / 4 * 4, 0x5A);  // stale bytes behind the padding
  M.words.resize(clean.size() / 4);
  memcpy(M.words.data(), clean.data(), clean.size());
  // ---- tables (BuildTables) ----
  HuffTables &H = M.H;
  memset(&H, 0, sizeof H);
  const int bpm = sc.blocks_per_mcu;
  for (int t = 0; t < 2; t++) {
    memcpy(H.vals[t], sc.dc_vals[t], 256);
    memcpy(H.vals[2 + t], sc.ac_vals[t], 256);
  }
  for (int k = 0; k < bpm; k++) {
    const int comp = sc.comp_of_block[k];
    H.dc_mask |= (uint32_t)(sc.dc_sel[comp] & 1) << k;
    H.ac_mask |= (uint32_t)(sc.ac_sel[comp] & 1) << k;
  }
  H.bpm = bpm;
  for (int t = 0; t < 4; t++) CodeRanges(t < 2 ? sc.dc_bits[t] : sc.ac_bits[t - 2], H.maxcode[t], H.valoff[t], &H.l2_first[t], &H.l2_size[t]);
  for (int t = 0; t < 4; t++) {
    for (int w = 0; w < (1 << kFastBits); w++) H.fast[t][w] = FastEntry(H, t, w);
    for (int j = 0; j < kL2Entries; j++) H.l2[t][j] = L2Entry(H, t, j);
  }
  SyncTables &S = M.S;
  memset(&S, 0, sizeof S);
  for (int t = 0; t < 4; t++)
    for (int w = 0; w < (1 << kFastBits); w++) S.t32[t][w] = SyncEntry(H, t, w);
  memcpy(S.l2, H.l2, sizeof S.l2); memcpy(S.l2_first, H.l2_first, sizeof S.l2_first);
  memcpy(S.l2_size, H.l2_size, sizeof S.l2_size); memcpy(S.maxcode, H.maxcode, sizeof S.maxcode);
  memcpy(S.valoff, H.valoff, sizeof S.valoff); memcpy(S.vals, H.vals, sizeof S.vals);
  S.dc_mask = H.dc_mask; S.ac_mask = H.ac_mask; S.bpm = bpm;
  // ---- geometry ----
  const int total_blocks = sc.mcus_x * sc.mcus_y * bpm;
  const int seg_lanes = P.seg_threads - P.warm_lanes, seg_bytes = seg_lanes * P.slice_bytes;
  const int nseg = (int)clean_len > seg_bytes ? ((int)clean_len + seg_bytes - 1) / seg_bytes : 1;
  const long long by_slices = (long long)seg_lanes * (P.slice_bytes * 8 / 2 + 32), by_blocks = (long long)total_blocks + 128;
  const int seg_cap = (int)std::min(by_slices, by_blocks);
  std::vector<SegRec> segs(nseg);
  // ---- SyncKernel, one "workgroup" per segment ----
  for (int seg = 0; seg < nseg; seg++) {
    SegRec &sr = segs[seg];
    if (seg > 0 && (long long)seg * seg_bytes >= (long long)clean_len) {
      sr.lanes.assign(seg_lanes, Lane());
      continue;
    }
    std::vector<Lane> ln(P.seg_threads);
    std::vector<uint64_t> state(P.seg_threads);
    for (int t = 0; t < P.seg_threads; t++) {
      ln[t] = MakeLane((long long)seg * seg_lanes + t - P.warm_lanes, M.total_bits, P);
      state[t] = Pack(DecodeState{ln[t].begin, 0, 0});
    }
    M.Relax(ln, state);
    const int total = M.WriteStarts(ln, P.warm_lanes, seg_lanes, sr.starts, seg_cap);
    sr.lanes.assign(ln.begin() + P.warm_lanes, ln.end());
    bool crossed = false;
    for (int t = P.warm_lanes; t < P.seg_threads; t++) crossed |= ln[t].active && ln[t].crossed;
    for (int t = P.warm_lanes; t < P.seg_threads; t++) {
      const bool next_has_data = t + 1 < P.seg_threads && ln[t].end < M.total_bits;
      if (ln[t].active && !next_has_data) { sr.out = ln[t].out; sr.nstart_total = total; sr.crossed = crossed; }
    }
    if (M.total_bits == 0) { sr.out = Pack(DecodeState{0, 0, 0}); sr.nstart_total = 0; }
  }
  // ---- PropagateKernel ----
  uint64_t truth = Pack(DecodeState{0, 0, 0});
  int block_base = 0;
  for (int seg = 0; seg < nseg; seg++) {
    SegRec &sr = segs[seg];
    if (seg > 0 && (long long)seg * seg_bytes >= (long long)clean_len) {
      sr.out = truth; sr.block_base = block_base;
      continue;
    }
    if (M.total_bits != 0 && sr.lanes[0].in != truth) {
      M.repairs++;
      std::vector<Lane> ln(P.seg_threads);
      std::vector<uint64_t> state(P.seg_threads);
      for (int t = 0; t < P.seg_threads; t++) {
        const bool mine = t < seg_lanes;
        ln[t] = MakeLane(mine ? (long long)seg * seg_lanes + t : -1, M.total_bits, P);
        state[t] = t == 0 ? truth : (mine ? sr.lanes[t].in : kNoState);
      }
      M.Relax(ln, state);
      const int total = M.WriteStarts(ln, 0, seg_lanes, sr.starts, seg_cap);
      bool crossed = false;
      for (int t = 0; t < seg_lanes; t++) crossed |= ln[t].active && ln[t].crossed;
      for (int t = 0; t < seg_lanes; t++) {
        sr.lanes[t] = ln[t];
        const bool next_has_data = t + 1 < seg_lanes && ln[t].end < M.total_bits;
        if (ln[t].active && !next_has_data) { sr.out = ln[t].out; sr.nstart_total = total; sr.crossed = crossed; }
      }
    }
    if (sr.crossed) return Fail("status 4: segment %d: a decode from the true state ran over a restart boundary", seg);
    sr.block_base = block_base;
    block_base += sr.nstart_total;
    truth = sr.out;
  }
  const int total_starts = block_base;
  if (total_starts - 1 < total_blocks) return Fail("status 2: %d block starts for %d blocks", total_starts, total_blocks);
  // ---- DcKernel ----
  std::vector<uint32_t> blk_pos(total_blocks, 0xFFFFFFFFu);
  std::vector<int32_t> blk_dc(total_blocks, 0);
  std::vector<uint16_t> blk_seg(total_blocks, 0xFFFF);
  for (int seg = 0; seg < nseg; seg++) {
    SegRec &sr = segs[seg];
    if (seg > 0 && (long long)seg * seg_bytes >= (long long)clean_len) continue;
    const int nstart = std::min(sr.nstart_total, seg_cap);
    int carry[3] = {0, 0, 0};
    for (int j = 0; j < nstart; j++) {
      const int ordinal = sr.block_base + j;
      if (!(ordinal < total_blocks && ordinal + 1 < total_starts)) continue;
      const int k = ordinal % bpm, comp = sc.comp_of_block[k];
      uint32_t used = 0;
      const uint32_t pos = sr.starts[j];
      if (pos == 0xFFFFFFFFu) return Fail("segment %d start %d was never written", seg, j);
      const int diff = DecodeDc(H, M.words.data(), pos, sc.dc_sel[comp] & 1, &used);
      carry[comp] += diff;
      blk_pos[ordinal] = pos + used;
      blk_dc[ordinal] = carry[comp];
      blk_seg[ordinal] = (uint16_t)seg;
    }
    for (int c = 0; c < 3; c++) sr.dc_total[c] = carry[c];
  }
  // ---- BlockKernel: workgroups of MCUs, tasks of 64 blocks of one class ----
  int mpw = (P.blocks_per_wg / bpm) / 32 * 32;
  if (mpw < 32) mpw = 32;
  const int total_mcus = total_blocks / bpm;
  uint8_t klist[12];
  int n0 = 0;
  for (int k = 0; k < bpm; k++) if ((sc.ac_sel[sc.comp_of_block[k]] & 1) == 0) klist[n0++] = (uint8_t)k;
  { int q = n0; for (int k = 0; k < bpm; k++) if ((sc.ac_sel[sc.comp_of_block[k]] & 1) != 0) klist[q++] = (uint8_t)k; }
  const int n1 = bpm - n0;
  std::vector<uint8_t> seen(total_blocks, 0);
  long long mismatches = 0;
  for (int m0 = 0; m0 < total_mcus; m0 += mpw) {
    const int Mm = std::min(mpw, total_mcus - m0);
    const int tasks0 = (Mm * n0 + 63) >> 6, tasks1 = (Mm * n1 + 63) >> 6;
    for (int task = 0; task < tasks0 + tasks1; task++) {
      const bool cls = task >= tasks0;
      const int ncls = cls ? n1 : n0;
      for (int lane = 0; lane < 64; lane++) {
        const int j = (cls ? task - tasks0 : task) * 64 + lane;
        const int mi = j / ncls;
        const int k = klist[(cls ? n0 : 0) + (j - mi * ncls)];
        const int mcu = m0 + mi, ordinal = mcu * bpm + k;
        const bool needed = mi < Mm && ordinal < total_blocks && ordinal + 1 < total_starts;
        if (!needed) continue;
        if (seen[ordinal]++) return Fail("block %d is decoded twice", ordinal);
        const int comp = sc.comp_of_block[k];
        int16_t coef[66];
        memset(coef, 0, sizeof coef);
        int dc = blk_dc[ordinal], seg0 = 0;
        const int interval_blocks = sc.restart_interval * bpm;
        if (interval_blocks) {
          const int first = ordinal / interval_blocks * interval_blocks;
          if (first > 0) {
            int klast = 0;
            for (int kk = 0; kk < bpm; kk++) if (sc.comp_of_block[kk] == comp) klast = kk;
            const int q = first - bpm + klast;
            dc -= blk_dc[q];
            seg0 = blk_seg[q];
          }
        }
        for (int s = seg0; s < blk_seg[ordinal]; s++) dc += segs[s].dc_total[comp];
        coef[0] = (int16_t)dc;
        if (blk_pos[ordinal] == 0xFFFFFFFFu) return Fail("block %d has no position", ordinal);
        DecodeBlockAc(H, M.words.data(), blk_pos[ordinal], 2u + (sc.ac_sel[comp] & 1), coef);
        // compare with the host decoder's column-major block
        const int my = mcu / sc.mcus_x, mx = mcu - my * sc.mcus_x;
        const int bx = mx * info.h_samp[comp] + sc.h_of_block[k], by = my * info.v_samp[comp] + sc.v_of_block[k];
        const int16_t *rb = ref[comp].data() + ((size_t)by * info.blocks_x[comp] + bx) * 64;
        for (int z = 0; z < 64; z++) {
          if (coef[z] != rb[kZigZagColMajorTable[z]]) {
            if (!mismatches)
              Fail("block %d (mcu %d, k %d, comp %d) coefficient z=%d: model %d, host decoder %d", ordinal, mcu, k, comp, z,
                   coef[z], rb[kZigZagColMajorTable[z]]);
            mismatches++;
          }
        }
      }
    }
  }
  for (int o = 0; o < total_blocks; o++)
    if (!seen[o]) return Fail("block %d is never decoded", o);
  if (stats) {
    stats[0] = nseg; stats[1] = M.rounds_max; stats[2] = M.repairs; stats[3] = M.overflow_lanes; stats[4] = total_blocks;
    stats[5] = total_starts; stats[6] = (int)M.rst_pos.size();
  }
  return mismatches ? 1 : 0;
}
