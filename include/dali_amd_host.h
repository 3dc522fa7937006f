/*
 * dali_amd_host.h -- C ABI of libdali_amd_host.so: the CPU-side pieces of the hot path that the
 * reference also keeps on the host (header parsing, entropy decode in the "hybrid" decoder,
 * random crop generation, argument preparation).  Pure host code (g++), no HIP.
 *
 * Conventions as in dali_amd_kernels.h: int status (0 = success), thread-local message via
 * daliamdHostGetLastErrorMessage(), caller owns all buffers, no exceptions across the ABI.
 */
#ifndef DALI_AMD_HOST_H_
#define DALI_AMD_HOST_H_

#include <stddef.h>
#include <stdint.h>

#include "dali_amd_kernels.h" /* descriptor PODs shared with the device library (plain C, no HIP) */

#ifdef __cplusplus
extern "C" {
#endif

#define DALIAMD_HOST_API __attribute__((visibility("default")))

DALIAMD_HOST_API const char *daliamdHostGetLastErrorMessage(void);

/* ----------------------------------------------------------------------------------------------
 * JPEG stream parsing + Huffman entropy decoding (baseline and progressive, 8-bit).
 * Replaces ImageDecoder::ParseSample (nvimgcodecCodeStreamGetImageInfo,
 * dali/operators/imgcodec/image_decoder.h:473-500) and the CPU half of the hybrid decode
 * nvImageCodec performs for device="mixed" (image_decoder.h:810-815).
 * -------------------------------------------------------------------------------------------- */
typedef struct {
  int32_t width, height;
  int32_t num_components;   /* 1, 3 or 4 (CMYK / YCCK: decoded on the host only)  */
  int32_t progressive;
  int32_t h_samp[4], v_samp[4];
  int32_t hmax, vmax;
  int32_t blocks_x[4], blocks_y[4]; /* allocated blocks per component (padded to the MCU) */
  int32_t down_w[4], down_h[4];     /* ceil(width*h/hmax), ceil(height*v/vmax) */
  int32_t orientation;              /* EXIF orientation 1..8 (1 when absent) */
  int32_t color;                    /* 0 gray, 1 YCbCr, 2 RGB (daliamdJpegColor_t); 4 components: 3 CMYK, 4 YCCK,
                                       + 8 when an Adobe marker is present (samples stored inverted) */
  int32_t restart_interval;
  int64_t coef_elems[4];            /* int16 elements of each component's coefficient array */
} daliamdJpegInfo;

DALIAMD_HOST_API int daliamdJpegParse(const uint8_t *data, size_t size, daliamdJpegInfo *info);

/* Entropy-decodes the stream into per-component coefficient arrays
 *   coef[c] : int16 [blocks_y][blocks_x][64], each block COLUMN-MAJOR (element = col*8 + row),
 *             un-dequantised; the arrays are cleared by the callee;
 *   quant   : uint16 [num_components][64] in the same column-major element order.
 * This is exactly the layout daliamdJpegIdctDesc consumes. */
DALIAMD_HOST_API int daliamdJpegDecodeCoefficients(const uint8_t *data, size_t size, const daliamdJpegInfo *info,
                                                  int16_t *const coef[4], uint16_t *quant);

/* The whole decode on the host: entropy decode + dequantisation + accurate integer IDCT + fancy (triangle) chroma
 * upsampling + BT.601 YCbCr -> RGB (gray replicated) + EXIF orientation `orientation` (0/1: as stored; 5..8: the
 * output is height x width turned, i.e. `out` holds info->width rows of info->height pixels).  out: u8 RGB rows of
 * `pitch` bytes.  This is decoders.image(device="cpu") - ImageDecoder<CPUBackend>, dali/operators/imgcodec/
 * image_decoder.h:613-880 / host_decoder.cc:35-48 over libjpeg-turbo; the same bytes the device path produces. */
DALIAMD_HOST_API int daliamdJpegDecodeRgbHost(const uint8_t *data, size_t size, const daliamdJpegInfo *info,
                                             int orientation, uint8_t *out, int64_t pitch);
/* The same with the decoder's `output_type` (values of DALIImageType, include/dali/core/common.h): RGB, BGR, GRAY
 * (one channel: the luma plane of a YCbCr / gray stream as libjpeg-turbo's JCS_GRAYSCALE output gives it, else
 * 0.299 R + 0.587 G + 0.114 B), YCbCr (ITU-R BT.601 with head room, as ConvertCPU: dali/operators/imgcodec/util/
 * convert.h:140-192), ANY_DATA (gray stays one channel, everything else RGB).  CMYK / YCCK streams (4 components)
 * become RGB with Pillow's formula (the reference's lives in un-vendored nvImageCodec; real ImageNet holds 22 such
 * files).  Output rows hold daliamdJpegOutputChannels(...) bytes per pixel. */
typedef enum {
  DALIAMD_IMAGE_RGB = 0, DALIAMD_IMAGE_BGR = 1, DALIAMD_IMAGE_GRAY = 2, DALIAMD_IMAGE_YCBCR = 3, DALIAMD_IMAGE_ANY = 4
} daliamdImageType;
DALIAMD_HOST_API int daliamdJpegOutputChannels(int num_components, int output_type);
DALIAMD_HOST_API int daliamdJpegDecodeHost(const uint8_t *data, size_t size, const daliamdJpegInfo *info, int orientation,
                                          int output_type, uint8_t *out, int64_t pitch);
/* RGB rows (e.g. from daliamdImageDecodeRgb) -> `output_type`; GRAY writes one byte per pixel. */
DALIAMD_HOST_API int daliamdConvertRgbRows(const uint8_t *rgb, int64_t in_pitch, int width, int height, int output_type,
                                          uint8_t *out, int64_t out_pitch);

/* Scan analysis for the GPU entropy decoder (libdali_amd_kernels: daliamdJpegHuffman*).  A stream is eligible
 * when it is baseline (SOF0/SOF1) and has ONE scan that interleaves all components (or is grayscale), with or without
 * restart intervals; everything else (progressive, multi-scan, more than two tables per class) goes through
 * daliamdJpegDecodeCoefficients. */
typedef struct {
  int32_t eligible;
  int32_t blocks_per_mcu, mcus_x, mcus_y;
  int64_t ecs_offset, ecs_length;     /* entropy-coded segment inside the stream, up to (excluding) the next marker */
  uint8_t comp_of_block[10];          /* component of each block of an MCU, in decode order */
  uint8_t h_of_block[10], v_of_block[10]; /* position of that block inside the component's MCU footprint */
  uint8_t dc_sel[4], ac_sel[4];       /* per component: Huffman table slot (0..3) */
  uint8_t dc_bits[4][16], dc_vals[4][256]; /* DHT contents per slot (class 0) */
  uint8_t ac_bits[4][16], ac_vals[4][256]; /* DHT contents per slot (class 1) */
  uint16_t quant[4][64];              /* per component, column-major element order (as daliamdJpegIdctDesc.quant) */
  int32_t restart_interval;           /* DRI: MCUs per restart interval, 0 = none (-> daliamdJpegHuffDesc.restart_interval) */
  int32_t length_is_upper_bound;      /* ecs_length runs to the end of the stream; the segment ends at the first marker
                                         that is not RSTn, which the GPU decoder's un-stuffing pass finds by itself */
} daliamdJpegScan;
/* Walks the scan: ecs_length is exact (one memchr pass over the entropy-coded bytes). */
DALIAMD_HOST_API int daliamdJpegAnalyzeScan(const uint8_t *data, size_t size, const daliamdJpegInfo *info,
                                           daliamdJpegScan *scan);
/* Header parse and scan analysis in ONE pass that stops at the SOS header (a few hundred bytes): fills `info` like
 * daliamdJpegParse and `scan` like daliamdJpegAnalyzeScan, except that ecs_length is "the rest of the stream"
 * (length_is_upper_bound = 1).  This is what decoders.image(device="mixed") runs per sample: the bytes of the scan are
 * first looked at on the device. */
DALIAMD_HOST_API int daliamdJpegAnalyzeHeader(const uint8_t *data, size_t size, daliamdJpegInfo *info,
                                             daliamdJpegScan *scan);

/* Indexed JPEG container (".didx"; dali_amd/host/jpeg_indexed.cpp, tools/jpeg2idx.py): a baseline JPEG prepared offline for
 * the GPU entropy decoder - its headers up to SOS as they are, its entropy-coded segment replaced by the index entry of
 * daliamdJpegHuffDesc.index (daliamdJpegHuffmanIndexBuildHost).  decoders.image(device="mixed") takes such samples wherever it
 * takes JPEG files and decodes them from the entry: the position passes of the decoder do not run, in the first epoch and in a
 * cold process too (the reference's offline indices: tools/tfrecord2idx, tools/wds2idx.py, tools/rec2idx.py).
 * Build: out == NULL returns the size in *length.  Returns 0 on success; a stream the GPU decoder does not take (progressive,
 * restart intervals, four components, truncated) is an error with a message - such files stay what they are. */
typedef struct {
  const uint8_t *header;   /* the JPEG's bytes up to and including its SOS header */
  int32_t header_len;
  int32_t ecs_len;         /* length of the original entropy-coded segment: daliamdJpegHuffDesc.ecs_len of the decode */
  int64_t index_offset;    /* offset of the index entry inside the container (a multiple of 64) */
  int64_t index_bytes;
  int64_t jpeg_size;       /* size of the file the container was made from */
} daliamdJpegIndexedView;
DALIAMD_HOST_API int daliamdJpegIndexedIs(const uint8_t *data, size_t size);
DALIAMD_HOST_API int daliamdJpegIndexedParse(const uint8_t *data, size_t size, daliamdJpegIndexedView *view);
/* What can be checked about a container's index entry without decoding (see jpeg_indexed.cpp): called by the mixed decoders with
 * the MCU structure the container's own headers announce, before anything is uploaded. */
DALIAMD_HOST_API int daliamdJpegIndexedValidate(const uint8_t *data, size_t size, const daliamdJpegIndexedView *view,
                                               int blocks_per_mcu, int total_blocks);
DALIAMD_HOST_API int daliamdJpegIndexedBuild(const uint8_t *jpeg, size_t size, uint8_t *out, size_t capacity, size_t *length);

/* Writes decoded coefficient arrays (the layout of daliamdJpegDecodeCoefficients) out again as ONE sequential, interleaved
 * baseline Huffman scan with the typical code tables of T.81 Annex K.3: the lossless re-encoding `jpegtran` performs.  `out`
 * receives the entropy-coded segment (byte-stuffed, no markers), `scan` the analysis the GPU entropy decoder needs for it
 * (eligible = 1, ecs_offset 0, exact ecs_length, the Annex K tables, `quant` as given).  decoders.image(mixed) uses it to
 * keep progressive / multi-scan streams resident in the encoded-stream cache in a form the device decodes (the reference
 * hands such streams to nvJPEG's hybrid decoder every epoch: image_decoder.h:810-834).  Returns 0 on success; coefficients
 * outside the baseline range (|AC| > 1023, |DC difference| > 2047) or a too small buffer are errors. */
DALIAMD_HOST_API int daliamdJpegEncodeBaselineScan(const daliamdJpegInfo *info, const int16_t *const coef[4], const uint16_t *quant,
                                                  uint8_t *out, size_t capacity, size_t *length, daliamdJpegScan *scan);

/* ----------------------------------------------------------------------------------------------
 * Random machinery, bit-compatible with the reference's host code.
 *   Philox4x32-10                 include/dali/core/random/philox.h:27-160
 *   RandomCropGenerator           dali/operators/image/crop/random_crop_generator_util.cc:36-101
 *   per-sample stream derivation  dali/operators/random/rng_base.h:95-140,
 *                                 dali/operators/image/crop/random_crop_attr.h:87-95
 *   coin_flip                     dali/operators/random/random_dist.h:293-312
 * -------------------------------------------------------------------------------------------- */
typedef struct {
  uint64_t key;
  uint64_t ctr[2]; /* [0] = low counter word pair, [1] = high ("sequence") */
  int32_t phase;
} daliamdPhiloxState;

/* One batch of RandomResizedCrop windows: sample i of iteration `master` uses
 * Philox(key ^ 0x12345678abcdefe, ctr_hi + i*65537, ctr_lo, phase).  After the call the caller
 * advances master.ctr[1] by the batch size (OperatorWithRng::Advance). */
DALIAMD_HOST_API int daliamdRandomCropBatch(const daliamdPhiloxState *master, int batch, const int32_t *shapes_hw,
                                           float aspect_lo, float aspect_hi, float area_lo, float area_hi,
                                           int num_attempts, int32_t *anchors_yx, int32_t *crops_hw);
DALIAMD_HOST_API int daliamdCoinFlipBatch(const daliamdPhiloxState *master, int batch, const float *probability,
                                         int probability_stride, int32_t *out);
DALIAMD_HOST_API void daliamdPhiloxAdvanceSequence(daliamdPhiloxState *state, uint64_t n);
DALIAMD_HOST_API int daliamdPhiloxStateToString(const daliamdPhiloxState *state, char *buf, int buf_len);
DALIAMD_HOST_API int daliamdPhiloxStateFromString(daliamdPhiloxState *state, const char *str);
/* Raw generator, for tests: fills out[0..n) with successive outputs and advances the state. */
DALIAMD_HOST_API void daliamdPhiloxGenerate(daliamdPhiloxState *state, uint32_t *out, int n);

/* CropMirrorNormalize argument preparation (double arithmetic, float result):
 * mean' = fma(-shift, std/scale, mean), inv_std = scale/std
 * (dali/operators/image/crop/crop_mirror_normalize.h:120-149).  Returns the number of entries
 * written (0 when normalisation is the identity and is skipped), or -1 on error. */
DALIAMD_HOST_API int daliamdCmnNormArgs(const float *mean, int nmean, const float *stddev, int nstd, float scale,
                                       float shift, float *mean_out, float *inv_std_out);
/* CropAttr::CalculateAnchor (dali/operators/image/crop/crop_attr.cc:224-240) */
DALIAMD_HOST_API int64_t daliamdCropAnchor(float anchor_norm, int64_t crop, int64_t in, int round_half_away);

/* ----------------------------------------------------------------------------------------------
 * CPU backend of the resampling / CropMirrorNormalize operators: ONE sample of a descriptor table that was filled for
 * the device kernels (daliamdResampleSetup; daliamdCmnDesc), executed on the calling thread with host pointers in
 * `in` / `out`.  Same arithmetic as the kernels, which follow the reference's CPU kernels
 * (dali/kernels/imgproc/resample/separable_cpu.h:152-241, resampling_impl_cpu.{h,cc},
 * dali/kernels/slice/slice_flip_normalize_permute_pad_cpu.h:37-64): RandomResizedCrop / Resize / CropMirrorNormalize
 * registered for CPU (random_resized_crop.cc:54, resize.cc, crop_mirror_normalize.cc:83) call these, one thread-pool
 * task per sample (resize_op_impl_cpu.h:84-107).
 * -------------------------------------------------------------------------------------------- */
DALIAMD_HOST_API int daliamdResampleRunHost(const daliamdResampleDesc *desc);
DALIAMD_HOST_API int daliamdCmnRunHost(const daliamdCmnDesc *desc);

/* The loss-less container formats decoders.image accepts next to JPEG ("Supported formats: JPEG, JPEG 2000, TIFF, PNG,
 * BMP, PNM, PPM, PGM, PBM, WebP", dali/operators/imgcodec/decoder_schema.cc:149), decoded on the host to 8-bit RGB
 * (gray replicated, alpha dropped, 16-bit samples reduced to their high byte).  Not implemented: TIFF, JPEG 2000, WebP.
 * daliamdImageProbe: format and - except for JPEG, see daliamdJpegParse - the image size, from the header only.
 * daliamdImageDecodeRgb: the window {y0, x0, h, w} of the image (h == w == 0: all of it) into rows of `pitch` bytes. */
typedef enum {
  DALIAMD_IMAGE_UNKNOWN = 0, DALIAMD_IMAGE_JPEG = 1, DALIAMD_IMAGE_PNG = 2, DALIAMD_IMAGE_BMP = 3, DALIAMD_IMAGE_PNM = 4
} daliamdImageFormat;
DALIAMD_HOST_API int daliamdImageProbe(const uint8_t *data, size_t size, daliamdImageFormat *format, int32_t *width,
                                      int32_t *height);
DALIAMD_HOST_API int daliamdImageDecodeRgb(const uint8_t *data, size_t size, uint8_t *out, int64_t pitch, int32_t y0,
                                          int32_t x0, int32_t h, int32_t w);

/* Bookkeeping of the decoded-image cache of decoders.image (`cache_size`, `cache_type`, `cache_threshold`): which
 * image is kept and at which offset of the one HBM blob.  Host only - the decoder writes the decoded image at
 * blob + offset and later hands out that address; no copy on either side.
 *   type "threshold": ImageCacheBlob::Add     (dali/operators/decoder/cache/image_cache_blob.cc:89-115)
 *   type "largest":   ImageCacheLargest::Add  (dali/operators/decoder/cache/image_cache_largest.cc:25-89)
 * Create returns NULL (and sets the host error message) for an unknown type or a threshold above the cache size.
 * OnDecode registers one decode of `key` (data_size = H*W*C is compared with the threshold; stored_size = bytes taken
 * in the blob) and returns the offset to store it at, or -1 when it is not kept.  Find: offset of a kept image or -1. */
DALIAMD_HOST_API void *daliamdImageCachePolicyCreate(const char *type, uint64_t cache_size, uint64_t threshold);
DALIAMD_HOST_API void daliamdImageCachePolicyDestroy(void *policy);
DALIAMD_HOST_API int64_t daliamdImageCachePolicyOnDecode(void *policy, const char *key, uint64_t data_size,
                                                         uint64_t stored_size);
DALIAMD_HOST_API int64_t daliamdImageCachePolicyFind(void *policy, const char *key);

/* Audio resampling on the host, the arithmetic of daliamdAudioResampleRun (windowed sinc, `quality` 0..100 -> 3..64
 * lobes): decoders.audio(sample_rate=...) and audio_resample(device="cpu")
 * (dali/operators/decoder/audio/audio_decoder_impl.cc:47-120, dali/kernels/signal/resampling_cpu.cc:129-236).
 * in: [in_length][channels] f32, out: [out_length][channels] f32.  Returns 0 on success. */
DALIAMD_HOST_API int daliamdAudioResampleHost(const float *in, int64_t in_length, int channels, double in_rate, double out_rate,
                                              float quality, float *out, int64_t out_length);

/* fn.normalize on the host (CPU backend of Normalize; the arithmetic of daliamdNormalizeRun:
 * dali/operators/math/normalize/normalize.cc:209-296, normalize_utils.h:133-220).  A sample is viewed as
 * [outer][reduced][inner], dense; the `n` samples of a call share their statistics (1, or the batch for batch=True).
 * in: uint8 or float; out: float, uint8 or int8 (daliamdDType_t).  has_stddev: scalar_inv_std = scale / stddev as the
 * operator folds it.  Returns 0 on success. */
typedef struct {
  const void *in;
  void *out;
  int64_t outer, reduced, inner;
} daliamdNormalizeHostSample;
DALIAMD_HOST_API int daliamdNormalizeHost(const daliamdNormalizeHostSample *samples, int n, int in_dtype, int out_dtype, int has_mean,
                                          float scalar_mean, int has_stddev, float scalar_inv_std, int ddof, float epsilon,
                                          float scale, float shift);

/* The heavy-augmentation operators on the host (CPU backend of WarpAffine / GaussianBlur / ColorTwist / Erase): one
 * sample per call, the descriptor of the device kernel with host pointers (the wg_* / tile_* fields are ignored), the
 * same arithmetic (dali/kernels/imgproc/warp_cpu.h:143-178, sampler.h:258-338, convolution/convolution_cpu.h:241-340,
 * pointwise/linear_transformation_cpu.h:57-77, erase/erase_cpu.h).  Return 0 on success. */
DALIAMD_HOST_API int daliamdWarpAffineHost(const daliamdWarpAffineDesc *desc);
DALIAMD_HOST_API int daliamdGaussianBlurHost(const daliamdGaussianBlurDesc *desc);
DALIAMD_HOST_API int daliamdPointwiseHost(const daliamdPointwiseDesc *desc);

/* The audio feature operators on the host (CPU backend of Spectrogram / MelFilterBank / ToDecibels / MFCC): the
 * arithmetic of the device kernels of include/dali_amd_kernels.h restated for one sample
 * (dali/kernels/signal/window/extract_windows_cpu.cc:96-145, fft/fft_cpu_impl_ffts.cc:105-111,
 * audio/mel_scale/mel_filter_bank_cpu.cc:77-111, signal/decibel/decibel_calculator.h:25-57, signal/dct/dct_cpu.cc:75-110).
 * spectrogram: out [nfft/2+1][num_windows] (num_windows from daliamdSpectrogramSetup); mel: weights [nfilter][nbins] from
 * daliamdMelFilterBankWeights; decibels: reference <= 0 takes the sample's maximum; dct: in [n_in][inner] -> out
 * [ndct][inner] with the table of daliamdDctTable and optional liftering coefficients.  Return 0 on success. */
DALIAMD_HOST_API int daliamdSpectrogramHost(const float *in, int64_t length, const daliamdSpectrogramParams *p, const float *window,
                                            int64_t num_windows, float *out);
DALIAMD_HOST_API int daliamdMelFilterBankHost(const float *spec, int nbins, int64_t frames, const float *weights, int nfilter,
                                              float *out);
DALIAMD_HOST_API int daliamdToDecibelsHost(const float *in, int64_t size, float multiplier, float reference, float cutoff_db,
                                           float *out);
DALIAMD_HOST_API int daliamdDctHost(const float *in, int n_in, int64_t inner, const float *table, const float *lifter, int ndct,
                                    float *out);

/* Normalised sample-type conversion on the host, the arithmetic of daliamdConvertNormRun (daliamdDType_t codes, mode 0 / 1 / 2
 * as there): the typed inputs and outputs of audio_resample(device="cpu") (dali/operators/audio/resample.cc:142-192,
 * include/dali/core/convert.h:262-350).  Returns 0 on success. */
DALIAMD_HOST_API int daliamdConvertNormHost(const void *in, int in_dtype, void *out, int out_dtype, int64_t count, int mode);

/* ----------------------------------------------------------------------------------------------
 * FLAC streams for decoders.audio (the reference decodes them through libsndfile / libFLAC:
 * dali/operators/decoder/audio/generic_decoder.cc:180-206; LibriSpeech ships as FLAC).  RFC 9639: constant, verbatim,
 * fixed-predictor and LPC subframes, partitioned Rice residuals with escapes, wasted bits, the three stereo
 * decorrelation modes, 4 to 32 bits per sample; CRC-8 / CRC-16 of every frame are checked.  Probe reads STREAMINFO
 * (frames = samples per channel; a stream that does not state it is walked once); Decode writes `frames` x channels
 * interleaved samples in the stream's own integer range.
 * -------------------------------------------------------------------------------------------- */
typedef struct {
  int32_t channels, bits_per_sample;
  double sample_rate;
  int64_t frames;
} daliamdAudioStreamInfo;
DALIAMD_HOST_API int daliamdFlacProbe(const uint8_t *data, size_t size, daliamdAudioStreamInfo *info);
DALIAMD_HOST_API int daliamdFlacDecode(const uint8_t *data, size_t size, int32_t *pcm, int64_t frames);

#ifdef __cplusplus
}
#endif
#endif /* DALI_AMD_HOST_H_ */
