// Indexed JPEG container (".didx", round 6): a baseline JPEG prepared OFFLINE for the GPU entropy decoder - its headers as
// they are, its entropy-coded segment as the INDEX ENTRY the decoder would build behind a first decode (un-stuffed stream +
// 12 bytes of decoder state per 256-byte slice; include/dali_amd_kernels.h: daliamdJpegHuffDesc.index).  A mixed decoder that
// receives such a sample uploads the entry where it would upload the segment and decodes from it: no un-stuffing, no
// relaxation, no hand-over check, no DC pass - in a cold process and in the first epoch, not only for streams that are
// already resident.  The reference prepares its containers offline the same way (tools/tfrecord2idx, tools/wds2idx.py,
// tools/rec2idx.py: index files that save every epoch the walk over the container).  tools/jpeg2idx.py writes them.
//
//   offset  0  "DAJX", u32 version (1), u32 header_len, u32 ecs_len, u64 index_bytes, u64 jpeg_size, zero up to 64
//   offset 64  the JPEG's bytes up to and including its SOS header (header_len), zero up to a multiple of 64
//   then       the index entry (index_bytes = daliamdJpegHuffmanIndexBytes(ecs_len); 64-byte aligned in the file)
#include <cstring>
#include <vector>

#include "dali_amd_host.h"
#include "dali_amd_kernels.h"
#include "host_common.h"

namespace {
constexpr char kMagic[4] = {'D', 'A', 'J', 'X'};
constexpr uint32_t kVersion = 1;
inline size_t Align64(size_t v) { return (v + 63) & ~(size_t)63; }
struct Head {
  char magic[4];
  uint32_t version, header_len, ecs_len;
  uint64_t index_bytes, jpeg_size;
};
static_assert(sizeof(Head) == 32, "layout");
}  // namespace

extern "C" {

int daliamdJpegIndexedIs(const uint8_t *data, size_t size) {
  return data && size >= 64 && memcmp(data, kMagic, 4) == 0;
}

int daliamdJpegIndexedParse(const uint8_t *data, size_t size, daliamdJpegIndexedView *view) {
  using namespace daliamd_host;
  if (!data || !view) return Fail("daliamdJpegIndexedParse: NULL argument");
  if (!daliamdJpegIndexedIs(data, size)) return Fail("not an indexed JPEG container");
  Head h;
  memcpy(&h, data, sizeof(h));
  if (h.version != kVersion) return Fail("indexed JPEG container of version %u (this build reads version %u)", h.version, kVersion);
  const size_t index_off = 64 + Align64(h.header_len);
  size_t want = 0;
  if (h.ecs_len > (1u << 30) || daliamdJpegHuffmanIndexBytes((int)h.ecs_len, &want) != DALIAMD_SUCCESS || want != h.index_bytes ||
      index_off + h.index_bytes > size || h.header_len < 4)
    return Fail("corrupt indexed JPEG container (header %u bytes, segment %u bytes, entry %llu bytes, file %zu bytes)", h.header_len,
                h.ecs_len, (unsigned long long)h.index_bytes, size);
  view->header = data + 64;
  view->header_len = h.header_len;
  view->ecs_len = (int32_t)h.ecs_len;
  view->index_offset = (int64_t)index_off;
  view->index_bytes = (int64_t)h.index_bytes;
  view->jpeg_size = (int64_t)h.jpeg_size;
  // the entry's own header must agree with the container (clean stream inside the room sized by ecs_len)
  int32_t ih[3];
  memcpy(ih, data + index_off, sizeof(ih));
  if (ih[0] < 0 || ih[0] > (int32_t)h.ecs_len || ih[1] < 0 || ih[2] != (ih[0] + 255) / 256)
    return Fail("corrupt indexed JPEG container (entry header %d / %d / %d)", ih[0], ih[1], ih[2]);
  return 0;
}

// The entry of a container comes from a FILE: before the device decodes from it, what can be checked without decoding is -
// a complete stream's worth of block starts, slice ordinals that start at zero, never decrease and stay inside the stream,
// block indices inside the MCU.  (What cannot: whether every slice really holds the blocks its ordinals promise.  The kernels
// do not trust that either - positions are clamped to the stream, blocks are only written inside the frame - so a forged entry
// yields wrong pixels or a decode error, never an access outside the decode's buffers: tools/fuzz_gpu_decoder.py FUZZ_CONTAINERS.)
int daliamdJpegIndexedValidate(const uint8_t *data, size_t size, const daliamdJpegIndexedView *view, int blocks_per_mcu,
                               int total_blocks) {
  using namespace daliamd_host;
  if (!data || !view || blocks_per_mcu < 1 || total_blocks < 0) return Fail("daliamdJpegIndexedValidate: invalid argument");
  if ((size_t)(view->index_offset + view->index_bytes) > size) return Fail("corrupt indexed JPEG container (entry outside the file)");
  const uint8_t *entry = data + view->index_offset;
  int32_t ih[3];
  memcpy(ih, entry, sizeof(ih));
  const int32_t clean_len = ih[0], total_starts = ih[1], num_slices = ih[2];
  if (clean_len < 0 || clean_len > view->ecs_len || num_slices != (clean_len + 255) / 256 ||
      total_starts < total_blocks + 1 || total_starts > total_blocks + 128)
    return Fail("corrupt indexed JPEG container (%d clean bytes, %d block starts for a frame of %d blocks)", clean_len, total_starts,
                total_blocks);
  const size_t entries_off = 64 + (((size_t)view->ecs_len + 256 + 63) & ~(size_t)63);
  const int cap = (view->ecs_len + 255) / 256;
  if (entries_off + 12 * (size_t)(cap + 1) > (size_t)view->index_bytes || num_slices > cap)
    return Fail("corrupt indexed JPEG container (slice entries outside the index)");
  uint32_t prev = 0;
  for (int s2 = 0; s2 <= cap; s2++) {
    uint32_t w[3];
    memcpy(w, entry + entries_off + 12 * (size_t)s2, 12);
    const uint32_t ordinal = w[0] & ((1u << 26) - 1u), c = (w[1] >> 12) & 15u, pos = w[1] & 4095u;
    if (s2 >= num_slices) {
      if (ordinal != (uint32_t)total_starts) return Fail("corrupt indexed JPEG container (slice %d behind the stream)", s2);
      continue;
    }
    if (ordinal < prev || ordinal > (uint32_t)total_starts || c >= (uint32_t)blocks_per_mcu || (s2 == 0 && (ordinal != 0 || pos != 0 || c != 0)))
      return Fail("corrupt indexed JPEG container (slice %d: ordinal %u after %u, block %u of %d)", s2, ordinal, prev, c, blocks_per_mcu);
    prev = ordinal;
  }
  return 0;
}

int daliamdJpegIndexedBuild(const uint8_t *jpeg, size_t size, uint8_t *out, size_t capacity, size_t *length) {
  using namespace daliamd_host;
  if (!jpeg || !length) return Fail("daliamdJpegIndexedBuild: NULL argument");
  *length = 0;
  daliamdJpegInfo info;
  daliamdJpegScan scan;
  if (daliamdJpegParse(jpeg, size, &info) != 0) return 1;
  if (daliamdJpegAnalyzeScan(jpeg, size, &info, &scan) != 0) return 1;
  if (!scan.eligible) return Fail("not a stream for the GPU entropy decoder (progressive, several scans, four components ...)");
  if (scan.restart_interval != 0) return Fail("streams with restart intervals have no index");
  daliamdJpegHuffDesc d{};
  d.ecs = jpeg + scan.ecs_offset;
  d.ecs_len = (int32_t)scan.ecs_length;
  d.blocks_per_mcu = scan.blocks_per_mcu;
  d.mcus_x = scan.mcus_x;
  d.total_blocks = scan.mcus_x * scan.mcus_y * scan.blocks_per_mcu;
  memcpy(d.comp_of_block, scan.comp_of_block, 10);
  memcpy(d.h_of_block, scan.h_of_block, 10);
  memcpy(d.v_of_block, scan.v_of_block, 10);
  memcpy(d.dc_sel, scan.dc_sel, 4);
  memcpy(d.ac_sel, scan.ac_sel, 4);
  for (int t = 0; t < 2; t++) {
    memcpy(d.bits[t], scan.dc_bits[t], 16);
    memcpy(d.bits[2 + t], scan.ac_bits[t], 16);
    memcpy(d.vals[t], scan.dc_vals[t], 256);
    memcpy(d.vals[2 + t], scan.ac_vals[t], 256);
  }
  size_t index_bytes = 0;
  if (daliamdJpegHuffmanIndexBytes(d.ecs_len, &index_bytes) != DALIAMD_SUCCESS) return Fail("%s", daliamdGetLastErrorMessage());
  const size_t header_len = (size_t)scan.ecs_offset, index_off = 64 + Align64(header_len), total = index_off + index_bytes;
  *length = total;
  if (!out) return 0;                                   // (size query)
  if (capacity < total) return Fail("daliamdJpegIndexedBuild: the output buffer holds %zu bytes, %zu are needed", capacity, total);
  memset(out, 0, index_off);
  Head h{};
  memcpy(h.magic, kMagic, 4);
  h.version = kVersion; h.header_len = (uint32_t)header_len; h.ecs_len = (uint32_t)d.ecs_len;
  h.index_bytes = index_bytes; h.jpeg_size = size;
  memcpy(out, &h, sizeof(h));
  memcpy(out + 64, jpeg, header_len);
  int32_t status = 0;
  if (daliamdJpegHuffmanIndexBuildHost(&d, out + index_off, &status) != DALIAMD_SUCCESS) return Fail("%s", daliamdGetLastErrorMessage());
  if (status != 0)
    return Fail(status == 3 ? "restart markers in a stream without a restart interval"
                            : "the entropy-coded segment ends before the last MCU (status %d)", status);
  return 0;
}

}  // extern "C"
