// Audio operators (BASELINE.json configs[3]): decoders.audio (host WAV parse), spectrogram, mel_filter_bank,
// to_decibels (device).  Arithmetic in libdali_amd_kernels.so.
//   decoders.audio   dali/operators/decoder/audio/audio_decoder_op.cc:25-179, generic_decoder.cc:140-220
//   Spectrogram      dali/operators/signal/fft/spectrogram.cc:30-87,152-296
//   MelFilterBank    dali/operators/audio/mel_scale/mel_filter_bank.cc:22-62
//   ToDecibels       dali/operators/signal/decibel/to_decibels_op.h:35-50, to_decibels_op_cpu.cc:22-47
#include <cmath>
#include <cstring>

#include "dali_amd_host.h"
#include "ops.h"
#include "pipeline.h"

namespace daliamd_host {

// =============================================================================================
// decoders.audio: RIFF/WAVE (8- / 16- / 24- / 32-bit PCM, 32-bit float) and FLAC, read the way libsndfile's
// sf_readf_short / sf_readf_int / sf_readf_float read them (generic_decoder.cc:170-183): integer samples of b bits
// become floats by 1 / 2^(b-1), int16 by keeping the top 16 bits (or shifting narrower ones up), int32 by shifting up
// =============================================================================================
DALI_SCHEMA(decoders__Audio)
    .DocStr("Decodes waveforms from encoded audio data.\n\nSupported in this build: RIFF/WAVE (8-, 16-, 24-, 32-bit PCM and "
            "32-bit float samples) and FLAC; Ogg/Vorbis is refused. The output is float32 in [-1, 1) (or int16 / int32, "
            "see `dtype`); the second output is the sampling rate.")
    .NumInput(1)
    .NumOutput(2)
    .AddOptionalArg("downmix", "If set to True, downmix all input channels to mono (1-D output).", ArgValue::Bool(false))
    .AddOptionalArg("dtype", "Output data type (FLOAT only).", ArgValue::Int(DALI_FLOAT))
    .AddOptionalTypeArg("sample_rate", "If specified, the target sample rate, in Hz, to which the audio is resampled.",
                        ArgType::FLOAT, true)
    .AddOptionalArg("quality", "Resampling quality, 0 is lowest, 100 is highest: 0 corresponds to 3 lobes of the sinc "
                    "filter; 50 gives 16 lobes and 100 gives 64 lobes.", ArgValue::Float(50.0));
DALI_SCHEMA(AudioDecoder).DocStr("Legacy alias of decoders.audio").NumInput(1).NumOutput(2).AddParent("decoders__Audio");

// tag: 1 = integer PCM in the file (`bits` wide, little endian; 8-bit is unsigned), 3 = float32, 0xF1AC = FLAC (decoded
// to int32 on demand)
struct WavInfo { int channels = 0, bits = 0, tag = 0; int64_t frames = 0; double rate = 0; const uint8_t *data = nullptr;
                 size_t nbytes = 0; };
enum { kTagFlac = 0xF1AC };

static WavInfo ParseWav(const uint8_t *p, size_t n, const std::string &src) {
  auto fail = [&](const char *why) { DALI_FAIL("Failed to decode ", src, ": ", why); };
  WavInfo w;
  if (n >= 4 && !memcmp(p, "fLaC", 4)) {
    daliamdAudioStreamInfo si;
    if (daliamdFlacProbe(p, n, &si) != 0) fail(daliamdHostGetLastErrorMessage());
    w.tag = kTagFlac; w.channels = si.channels; w.bits = si.bits_per_sample; w.rate = si.sample_rate; w.frames = si.frames;
    w.data = p; w.nbytes = n;
    return w;
  }
  if (n >= 4 && !memcmp(p, "OggS", 4)) fail("Ogg streams are not supported by this build (WAV and FLAC are)");
  if (n < 12 || memcmp(p, "RIFF", 4) || memcmp(p + 8, "WAVE", 4)) fail("not a RIFF/WAVE or FLAC stream");
  size_t pos = 12;
  bool have_fmt = false;
  while (pos + 8 <= n) {
    uint32_t size;
    memcpy(&size, p + pos + 4, 4);
    const uint8_t *body = p + pos + 8;
    if (!memcmp(p + pos, "fmt ", 4)) {
      if (size < 16 || pos + 8 + size > n) fail("truncated fmt chunk");
      uint16_t tag, ch, bits;
      uint32_t rate;
      memcpy(&tag, body, 2); memcpy(&ch, body + 2, 2); memcpy(&rate, body + 4, 4); memcpy(&bits, body + 14, 2);
      if (tag == 0xFFFE && size >= 26) memcpy(&tag, body + 24, 2);  // WAVE_FORMAT_EXTENSIBLE sub-format
      w.tag = tag; w.channels = ch; w.rate = rate; w.bits = bits;
      have_fmt = true;
    } else if (!memcmp(p + pos, "data", 4)) {
      if (!have_fmt) fail("data chunk before fmt chunk");
      size_t avail = std::min<size_t>(size, n - (pos + 8));
      if (!((w.tag == 1 && (w.bits == 8 || w.bits == 16 || w.bits == 24 || w.bits == 32)) || (w.tag == 3 && w.bits == 32)))
        fail("unsupported sample format (8- / 16- / 24- / 32-bit PCM and 32-bit float are supported)");
      if (w.channels < 1) fail("no channels");
      w.frames = (int64_t)(avail / (w.bits / 8) / w.channels);
      w.data = body;
      return w;
    }
    pos += 8 + size + (size & 1);
  }
  fail("no data chunk");
  return w;
}

// 16-bit little-endian PCM (at any byte alignment: the data chunk starts wherever the header ends) -> float, value *
// scale.  AVX2 clone where the CPU has it (8 samples per instruction), same arithmetic: one conversion, one product.
__attribute__((target_clones("avx2", "default")))
static void Pcm16ToFloat(const uint8_t *__restrict src, float *__restrict dst, int64_t count, float scale) {
  for (int64_t k = 0; k < count; k++) {
    int16_t v;
    memcpy(&v, src + 2 * k, 2);
    dst[k] = (float)v * scale;
  }
}

class AudioDecoderCpu : public OperatorBase {
 public:
  explicit AudioDecoderCpu(const OpSpec &spec) : OperatorBase(spec), downmix_(spec.GetBool("downmix")) {
    dtype_ = (DALIDataType)spec.GetInt("dtype");
    DALI_ENFORCE(dtype_ == DALI_FLOAT || dtype_ == DALI_INT16 || dtype_ == DALI_INT32,
                 "Unsupported output type: ", (int)dtype_, "\nSupported types are : int16, int32, float");   // audio_decoder_op.cc
    quality_ = (float)spec.GetFloat("quality");
    DALI_ENFORCE(quality_ >= 0 && quality_ <= 100, "Resampling quality must be in [0..100] range");
  }
  bool SetupImpl(std::vector<OutputDesc> &desc, const Workspace &ws) override {
    const TensorList &in = ws.Input(0);
    int n = in.num_samples();
    infos_.resize(n);
    desc[0].type = dtype_; desc[1].type = DALI_FLOAT;
    desc[0].shape.resize(n); desc[1].shape.assign(n, TensorShape{});
    for (int i = 0; i < n; i++) {
      std::string src = i < (int)in.source_info.size() && !in.source_info[i].empty() ? in.source_info[i] : make_string("sample #", i);
      infos_[i] = ParseWav(static_cast<const uint8_t *>(in.raw(i)), in.nbytes(i), src);
    }
    target_rate_.assign(n, 0.0f);
    if (spec_.ArgumentDefined("sample_rate")) target_rate_ = GetPerSampleFloat(spec_, ws, "sample_rate", n);
    // Graph-level fusion with a GPU Spectrogram behind the copy to the device (EmitPcm16): a batch of single-channel
    // 16-bit streams that needs no resampling leaves as the int16 samples the files hold - half the bytes through the
    // host's memory and over the bus, and no conversion pass here; the spectrogram kernel's load divides by 32768.
    batch_pcm16_ = emit_pcm16_ && dtype_ == DALI_FLOAT && n > 0;
    for (int i = 0; i < n && batch_pcm16_; i++) {
      const WavInfo &w = infos_[i];
      const bool resample = target_rate_[i] > 0 && (float)w.rate != target_rate_[i];
      batch_pcm16_ = !resample && w.channels == 1 && w.bits == 16 && (w.tag == 1 || w.tag == kTagFlac);
    }
    if (batch_pcm16_) desc[0].type = DALI_INT16;
    // ... and when every stream is a WAV file the samples are where the reader put them: the output is a VIEW of the
    // input (sample i = the data chunk of file i), nothing is copied on the host at all
    view_ = batch_pcm16_;
    for (int i = 0; i < n && view_; i++) view_ = infos_[i].tag == 1 && (reinterpret_cast<uintptr_t>(infos_[i].data) & 1) == 0;
    for (int i = 0; i < n; i++) {
      // DecodedAudioShape (audio_decoder_impl.cc:38-47)
      const bool resample = target_rate_[i] > 0 && (float)infos_[i].rate != target_rate_[i];
      int64_t len = infos_[i].frames;
      if (resample) len = (int64_t)std::ceil(infos_[i].frames * (double)target_rate_[i] / infos_[i].rate);
      if (downmix_ || infos_[i].channels == 1) desc[0].shape[i] = {len};
      else desc[0].shape[i] = {len, infos_[i].channels};
    }
    if (view_) {
      view_shapes_ = desc[0].shape;
      return false;   // RunImpl sizes the outputs itself
    }
    return true;
  }
  void RunImpl(Workspace &ws) override {
    TensorList &out = ws.Output(0), &rate = ws.Output(1);
    if (view_) {
      const int n = (int)infos_.size();
      std::vector<void *> ptrs(n);
      for (int i = 0; i < n; i++) ptrs[i] = const_cast<uint8_t *>(infos_[i].data);
      // (the input list keeps the reader's block alive as long as this view exists: the ring slot of the same iteration)
      out.Resize(view_shapes_, DALI_INT16, 1, ptrs, std::vector<int64_t>(n, 0), ws.inputs[0]);
      out.source_info = ws.Input(0).source_info;
      rate.Resize(std::vector<TensorShape>(n, TensorShape{}), DALI_FLOAT);
      for (int i = 0; i < n; i++) *static_cast<float *>(rate.raw(i)) = (float)infos_[i].rate;
      return;
    }
    for (int i = 0; i < out.num_samples(); i++) {
      ws.GetThreadPool().AddWork([&, i](int) {
        const WavInfo &w = infos_[i];
        bool mono_out = downmix_ || w.channels == 1;
        const bool resample = target_rate_[i] > 0 && (float)w.rate != target_rate_[i];
        // DecodeAudio<T> (audio_decoder_impl.cc:49-120): without resampling and downmixing the frames are decoded
        // straight to the output type (libsndfile's reads: PCM16 -> int16 as stored, -> int32 shifted up by 16 bits,
        // -> float divided by 32768; float32 files are read as floats only).  Otherwise the frames are decoded to
        // floats, downmixed with equal weights (kernels/signal/downmixing.h:50-76: the sum of sample * (1 / channels),
        // accumulated in channel order) and / or resampled, and the LAST of these steps converts to the output type
        // with ConvertSatNorm.
        const int och = mono_out ? 1 : w.channels;
        const bool downmix = w.channels > 1 && downmix_;
        const int kout = dtype_ == DALI_INT16 ? DALIAMD_INT16 : dtype_ == DALI_INT32 ? DALIAMD_INT32 : DALIAMD_FLOAT;
        if (dtype_ != DALI_FLOAT) DALI_ENFORCE(w.tag != 3, "decoders.audio: integer output from a float32 file is not supported");
        // integer samples in the stream's own range (`bits` wide): straight from the WAV bytes, or a decoded FLAC stream
        std::vector<int32_t> flac;
        if (w.tag == kTagFlac) {
          flac.resize((size_t)w.frames * w.channels);
          if (daliamdFlacDecode(w.data, w.nbytes, flac.data(), w.frames) != 0)
            DALI_FAIL("Failed to decode ", i < (int)ws.Input(0).source_info.size() ? ws.Input(0).source_info[i] : make_string("sample #", i),
                      ": ", daliamdHostGetLastErrorMessage());
        }
        if (batch_pcm16_) {   // (SetupImpl: mono, 16 bits, no resampling)
          int16_t *o16 = static_cast<int16_t *>(out.raw(i));
          if (w.tag == 1) memcpy(o16, w.data, (size_t)w.frames * 2);
          else for (int64_t k = 0; k < w.frames; k++) o16[k] = (int16_t)flac[k];
          *static_cast<float *>(rate.raw(i)) = (float)w.rate;
          return;
        }
        const int bits = w.bits;
        auto isample = [&](int64_t k) -> int32_t {
          if (w.tag == kTagFlac) return flac[k];
          const uint8_t *q = w.data + k * (bits / 8);
          switch (bits) {
            case 8: return (int32_t)q[0] - 128;   // 8-bit WAV is unsigned
            case 16: { int16_t sv; memcpy(&sv, q, 2); return sv; }
            case 24: return (int32_t)((uint32_t)q[0] << 8 | (uint32_t)q[1] << 16 | (uint32_t)q[2] << 24) >> 8;
            default: { int32_t sv; memcpy(&sv, q, 4); return sv; }
          }
        };
        const float to_float = 1.0f / (float)((int64_t)1 << (bits - 1));
        if (!resample && !downmix && dtype_ != DALI_FLOAT) {
          const int64_t count = w.frames * w.channels;
          if (dtype_ == DALI_INT16) {
            int16_t *o16 = static_cast<int16_t *>(out.raw(i));
            if (w.tag == 1 && bits == 16) memcpy(o16, w.data, (size_t)count * 2);
            else for (int64_t k = 0; k < count; k++) o16[k] = (int16_t)(bits > 16 ? isample(k) >> (bits - 16) : isample(k) * (1 << (16 - bits)));
          } else {
            int32_t *o32 = static_cast<int32_t *>(out.raw(i));
            for (int64_t k = 0; k < count; k++) o32[k] = (int32_t)((uint32_t)isample(k) << (32 - bits));
          }
          *static_cast<float *>(rate.raw(i)) = (float)w.rate;
          return;
        }
        if (dtype_ == DALI_FLOAT && !resample && !downmix && w.tag == 1 && bits == 16) {
          // the common case (16-bit PCM files, float output, nothing to mix): one pass the compiler vectorises - the
          // general loop below picks the sample width per sample (0.35 ms per 12 s utterance against 0.03 ms)
          Pcm16ToFloat(w.data, static_cast<float *>(out.raw(i)), w.frames * w.channels, to_float);
          *static_cast<float *>(rate.raw(i)) = (float)w.rate;
          return;
        }
        std::vector<float> scratch, mixed;
        const bool float_direct = dtype_ == DALI_FLOAT && !resample;   // the float result of decode / downmix IS the output
        if (!float_direct) scratch.resize((size_t)w.frames * och);
        float *o = float_direct ? static_cast<float *>(out.raw(i)) : scratch.data();
        const float weight = 1.0f / w.channels;
        for (int64_t f = 0; f < w.frames; f++) {
          float acc = 0;
          for (int c = 0; c < w.channels; c++) {
            float v;
            if (w.tag != 3) v = (float)isample(f * w.channels + c) * to_float;
            else memcpy(&v, w.data + (f * w.channels + c) * 4, 4);
            if (!mono_out) o[f * w.channels + c] = v;
            else if (w.channels == 1) acc = v;
            else if (c == 0) acc = v * weight;
            else acc += v * weight;
          }
          if (mono_out) o[f] = acc;
        }
        if (resample) {
          const int64_t out_len = out.shape(i)[0];
          std::vector<float> res;
          float *dst = static_cast<float *>(out.raw(i));
          if (dtype_ != DALI_FLOAT) { res.resize((size_t)out_len * och); dst = res.data(); }
          if (daliamdAudioResampleHost(scratch.data(), w.frames, och, w.rate, target_rate_[i], quality_, dst, out_len) != 0)
            DALI_FAIL(daliamdHostGetLastErrorMessage());
          if (dtype_ != DALI_FLOAT &&
              daliamdConvertNormHost(res.data(), DALIAMD_FLOAT, out.raw(i), kout, (int64_t)res.size(), 0) != 0)
            DALI_FAIL(daliamdHostGetLastErrorMessage());
        } else if (dtype_ != DALI_FLOAT) {   // downmix only: ConvertSatNorm<Out>(sum)
          if (daliamdConvertNormHost(scratch.data(), DALIAMD_FLOAT, out.raw(i), kout, (int64_t)scratch.size(), 0) != 0)
            DALI_FAIL(daliamdHostGetLastErrorMessage());
        }
        *static_cast<float *>(rate.raw(i)) = resample ? target_rate_[i] : (float)w.rate;
      }, infos_[i].frames);
    }
    ws.GetThreadPool().RunAll();
  }

  // decoders.audio -> copy to the device -> Spectrogram (gpu), each the only consumer of the one before
  bool CanEmitPcm16() const { return dtype_ == DALI_FLOAT; }
  void EmitPcm16() { emit_pcm16_ = true; }

 private:
  bool downmix_;
  bool emit_pcm16_ = false, batch_pcm16_ = false, view_ = false;
  std::vector<TensorShape> view_shapes_;
  float quality_ = 50.0f;
  DALIDataType dtype_ = DALI_FLOAT;
  std::vector<float> target_rate_;
  std::vector<WavInfo> infos_;
};
DALI_REGISTER_OPERATOR(decoders__Audio, AudioDecoderCpu, CPU);
DALI_REGISTER_OPERATOR(AudioDecoder, AudioDecoderCpu, CPU);

// =============================================================================================
// Spectrogram
// =============================================================================================
DALI_SCHEMA(Spectrogram)
    .DocStr("Produces a spectrogram from a 1D signal (for example, audio).\n\nInput data is expected to be one channel "
            "float32. The output is laid out frequency-major (\"ft\"): (nfft/2+1, number of windows).")
    .NumInput(1)
    .NumOutput(1)
    .AddOptionalTypeArg("nfft", "Size of the FFT (a power of two here). Default: window_length.", ArgType::INT)
    .AddOptionalArg("window_length", "Window size in number of samples.", ArgValue::Int(512))
    .AddOptionalArg("window_step", "Step between the STFT windows in number of samples.", ArgValue::Int(256))
    .AddOptionalTypeArg("window_fn", "Samples of the window function (default: Hann).", ArgType::FLOAT_VEC)
    .AddOptionalArg("power", "Exponent of the magnitude of the spectrum: 1 (amplitude) or 2 (power).", ArgValue::Int(2))
    .AddOptionalArg("center_windows", "Indicates whether extracted windows should be padded so that the window function is "
                    "centered at multiples of window_step.", ArgValue::Bool(true))
    .AddOptionalArg("reflect_padding", "Indicates the padding policy when sampling outside the bounds of the signal.",
                    ArgValue::Bool(true))
    .AddOptionalArg("layout", "Output layout: \"ft\" (frequency-major; the only one supported here).", ArgValue::Str("ft"));

class SpectrogramGpu : public OperatorBase {
 public:
  explicit SpectrogramGpu(const OpSpec &spec) : OperatorBase(spec), window_dev_(StorageDevice::GPU) {
    p_.window_length = (int)spec.GetInt("window_length");
    p_.nfft = spec.TryArg("nfft") ? (int)spec.GetInt("nfft") : p_.window_length;
    p_.window_step = (int)spec.GetInt("window_step");
    p_.power = (int)spec.GetInt("power");
    p_.center_windows = spec.GetBool("center_windows");
    p_.reflect_padding = spec.GetBool("reflect_padding");
    DALI_ENFORCE(spec.GetString("layout") == "ft", "Spectrogram (gpu): only layout=\"ft\" is supported");
    window_.resize(p_.window_length);
    if (spec.TryArg("window_fn")) {
      auto w = spec.GetFloatVec("window_fn");
      DALI_ENFORCE((int)w.size() == p_.window_length, "Window function should match the specified `window_length`");
      for (int i = 0; i < p_.window_length; i++) window_[i] = (float)w[i];
    } else {
      daliamdHannWindow(p_.window_length, window_.data());
    }
  }
  bool SetupImpl(std::vector<OutputDesc> &desc, const Workspace &ws) override {
    const TensorList &in = ws.Input(0);
    DALI_ENFORCE(in.type() == DALI_FLOAT || (accept_pcm16_ && in.type() == DALI_INT16),
                 "Spectrogram expects float32 input, got ", TypeName(in.type()));
    p_.input_pcm16 = in.type() == DALI_INT16;   // (only from a decoders.audio fused in front: see AcceptPcm16)
    int n = in.num_samples();
    descs_.assign(n, daliamdSpectrogramDesc{});
    for (int i = 0; i < n; i++) {
      DALI_ENFORCE(in.shape(i).size() == 1, "Spectrogram expects a 1-D (single channel) signal");
      descs_[i].in = static_cast<const float *>(in.raw(i));
      descs_[i].length = in.shape(i)[0];
    }
    KCHECK(daliamdSpectrogramSetup(descs_.data(), n, &p_, &nwg_, &lds_));
    desc[0].type = DALI_FLOAT;
    desc[0].shape.resize(n);
    for (int i = 0; i < n; i++) desc[0].shape[i] = {p_.nfft / 2 + 1, descs_[i].num_windows};
    return !(fused_ && ws.backend != OpType::CPU);   // fused: no buffer, the mel filter bank behind launches for both
  }
  void AcceptPcm16() { accept_pcm16_ = true; }
  // MelFilterBank is the only consumer: hand the arguments on (nfft 512 / 1024: the sizes the fused kernel exists for)
  bool EnableFusion() {
    fused_ = p_.nfft == 512 || p_.nfft == 1024;
    return fused_;
  }
  void RunImpl(Workspace &ws) override {
    TensorList &out = ws.Output(0);
    out.SetLayout("ft");
    int n = (int)descs_.size();
    if (!n) return;
    if (ws.backend == OpType::CPU) {  // one thread-pool task per sample on the host kernel
      for (int i = 0; i < n; i++) {
        float *dst = static_cast<float *>(out.raw(i));
        ws.GetThreadPool().AddWork([this, i, dst](int) {
          const auto &d = descs_[i];
          if (daliamdSpectrogramHost(d.in, d.length, &p_, window_.data(), d.num_windows, dst) != 0)
            DALI_FAIL(daliamdHostGetLastErrorMessage());
        }, descs_[i].length);
      }
      ws.GetThreadPool().RunAll();
      NoteLaunch(ws, "host_spectrogram");
      return;
    }
    if (!window_uploaded_) {
      std::vector<float> tables((window_.size() + 3) / 4 * 4 + p_.nfft);   // window, then the FFT twiddles (16-byte aligned)
      std::copy(window_.begin(), window_.end(), tables.begin());
      twiddle_offset_ = (window_.size() + 3) / 4 * 4;
      daliamdSpectrogramTwiddles(p_.nfft, tables.data() + twiddle_offset_);
      window_dev_.Reserve(tables.size() * sizeof(float));
      KCHECK(daliamdMemcpyH2DAsync(window_dev_.data(), tables.data(), tables.size() * sizeof(float), ws.stream));
      // one-time upload shared by all later iterations, which run on other streams: make it visible to them
      KCHECK(daliamdStreamSynchronize(ws.stream));
      KCHECK(daliamdStreamSynchronize(ws.stream));  // one-time: `window_` is pageable host memory
      window_uploaded_ = true;
    }
    const float *tables = static_cast<const float *>(window_dev_.data());
    if (fused_) {
      auto d = std::make_shared<DeferredAudio>();
      d->source = ws.inputs[0];
      d->params = p_;
      d->window_dev = tables;
      d->twiddles_dev = tables + twiddle_offset_;
      d->descs = descs_;
      d->nwg = nwg_;
      out.Resize({}, DALI_FLOAT);
      out.deferred_audio = d;
      return;
    }
    for (int i = 0; i < n; i++) descs_[i].out = static_cast<float *>(out.raw(i));
    auto *dev = static_cast<const daliamdSpectrogramDesc *>(uploader_.Upload(descs_.data(), n * sizeof(descs_[0]), ws.stream, ws.ring + 1));
    KCHECK(daliamdSpectrogramRun(ws.stream, dev, n, &p_, tables, tables + twiddle_offset_, nwg_, lds_));
    NoteLaunch(ws, "spectrogram");
  }

 private:
  daliamdSpectrogramParams p_{};
  bool accept_pcm16_ = false;
  std::vector<float> window_;
  Buffer window_dev_;
  bool window_uploaded_ = false;
  size_t twiddle_offset_ = 0;
  int nwg_ = 0, lds_ = 0;
  bool fused_ = false;
  std::vector<daliamdSpectrogramDesc> descs_;
  DescUploader uploader_;
};
DALI_REGISTER_OPERATOR(Spectrogram, SpectrogramGpu, GPU);
DALI_REGISTER_OPERATOR(Spectrogram, SpectrogramGpu, CPU);  // same class: the host kernel when run on the CPU

// =============================================================================================
// MelFilterBank
// =============================================================================================
DALI_SCHEMA(MelFilterBank)
    .DocStr("Converts a spectrogram to a mel spectrogram by applying a bank of triangular filters.\n\nThe frequency ('f') "
            "dimension is the first one (\"ft\" layout). Implemented as a dense filterbank x frames matrix product on the "
            "MI355X f32 matrix cores.")
    .NumInput(1)
    .NumOutput(1)
    .AddOptionalArg("nfilter", "Number of mel filters.", ArgValue::Int(128))
    .AddOptionalArg("sample_rate", "Sampling rate of the audio signal.", ArgValue::Float(44100.0))
    .AddOptionalArg("freq_low", "The minimum frequency.", ArgValue::Float(0.0))
    .AddOptionalArg("freq_high", "The maximum frequency. If not provided, sample_rate / 2 is used.", ArgValue::Float(0.0))
    .AddOptionalArg("normalize", "Normalize the triangular filter weights by the width of their frequency bands.",
                    ArgValue::Bool(true))
    .AddOptionalArg("mel_formula", "\"slaney\" or \"htk\".", ArgValue::Str("slaney"));

class MelFilterBankGpu : public OperatorBase {
 public:
  explicit MelFilterBankGpu(const OpSpec &spec) : OperatorBase(spec), weights_dev_(StorageDevice::GPU) {
    nfilter_ = (int)spec.GetInt("nfilter");
    sample_rate_ = (float)spec.GetFloat("sample_rate");
    freq_low_ = (float)spec.GetFloat("freq_low");
    freq_high_ = (float)spec.GetFloat("freq_high");
    normalize_ = spec.GetBool("normalize");
    std::string f = spec.GetString("mel_formula");
    DALI_ENFORCE(f == "slaney" || f == "htk", "Unsupported mel_formula value \"", f, "\". Supported values are: \"slaney\", \"htk\"");
    formula_ = f == "htk";
    DALI_ENFORCE(nfilter_ > 0, "`nfilter` must be positive");
  }
  void ExpectFusedInput() { fused_input_ = true; }
  bool HasFusedInput() const { return fused_input_; }
  void DeferToDecibels() { defer_ = true; }
  bool SetupImpl(std::vector<OutputDesc> &desc, const Workspace &ws) override {
    const TensorList &in = ws.Input(0);
    if (fused_input_ && ws.backend != OpType::CPU) {
      const DeferredAudio *def = in.deferred_audio.get();
      DALI_ENFORCE(def, "internal: MelFilterBank expected the deferred arguments of the Spectrogram in front of it");
      const int n = (int)def->descs.size(), nb = def->params.nfft / 2 + 1;
      DALI_ENFORCE(nbins_ == 0 || nb == nbins_, "All spectrograms must have the same number of frequency bins");
      nbins_ = nb;
      desc[0].type = DALI_FLOAT;
      desc[0].shape.resize(n);
      for (int i = 0; i < n; i++) desc[0].shape[i] = {nfilter_, def->descs[i].num_windows};
      return !defer_;
    }
    DALI_ENFORCE(in.type() == DALI_FLOAT, "MelFilterBank expects float32 input");
    int n = in.num_samples();
    descs_.assign(n, daliamdMelDesc{});
    desc[0].type = DALI_FLOAT;
    desc[0].shape.resize(n);
    for (int i = 0; i < n; i++) {
      DALI_ENFORCE(in.shape(i).size() == 2, "MelFilterBank expects a 2-D (frequency, time) spectrogram");
      int nb = (int)in.shape(i)[0];
      DALI_ENFORCE(nbins_ == 0 || nb == nbins_, "All spectrograms must have the same number of frequency bins (got ", nb,
                   " after ", nbins_, ")");
      nbins_ = nb;
      descs_[i].in = static_cast<const float *>(in.raw(i));
      descs_[i].frames = (int)in.shape(i)[1];
      desc[0].shape[i] = {nfilter_, in.shape(i)[1]};
    }
    return true;
  }
  void RunImpl(Workspace &ws) override {
    TensorList &out = ws.Output(0);
    out.SetLayout("ft");
    int n = (int)descs_.size();
    if (ws.backend == OpType::CPU) {
      if (!n) return;
      if (host_weights_.empty()) {
        host_weights_.resize((size_t)nfilter_ * nbins_);
        KCHECK(daliamdMelFilterBankWeights(nfilter_, 2 * (nbins_ - 1), sample_rate_, freq_low_, freq_high_, normalize_, formula_,
                                           host_weights_.data()));
      }
      for (int i = 0; i < n; i++) {
        float *dst = static_cast<float *>(out.raw(i));
        ws.GetThreadPool().AddWork([this, i, dst](int) {
          if (daliamdMelFilterBankHost(descs_[i].in, nbins_, descs_[i].frames, host_weights_.data(), nfilter_, dst) != 0)
            DALI_FAIL(daliamdHostGetLastErrorMessage());
        }, descs_[i].frames);
      }
      ws.GetThreadPool().RunAll();
      NoteLaunch(ws, "host_mel_filter_bank");
      return;
    }
    const DeferredAudio *def = fused_input_ ? ws.Input(0).deferred_audio.get() : nullptr;
    if (def) n = (int)def->descs.size();
    if (!n) return;
    if (weights_.empty()) {
      weights_.resize((size_t)nfilter_ * nbins_);
      KCHECK(daliamdMelFilterBankWeights(nfilter_, 2 * (nbins_ - 1), sample_rate_, freq_low_, freq_high_, normalize_, formula_,
                                         weights_.data()));
      std::vector<int32_t> bands(2 * (size_t)nfilter_);
      KCHECK(daliamdMelFilterBankBands(weights_.data(), nfilter_, nbins_, bands.data()));
      // the same filters as 16 x 4 tiles for the matrix cores (the fused kernel)
      int ntiles = 0;
      const int nrb = (nfilter_ + 15) / 16;
      KCHECK(daliamdMelFilterBankMfmaLayout(weights_.data(), nfilter_, nbins_, nullptr, nullptr, &ntiles));
      std::vector<float> tiles((size_t)ntiles * 64);
      std::vector<int32_t> row_blocks(4 * (size_t)nrb);
      KCHECK(daliamdMelFilterBankMfmaLayout(weights_.data(), nfilter_, nbins_, tiles.data(), row_blocks.data(), &ntiles));
      auto pad16 = [](size_t b) { return (b + 15) / 16 * 16; };
      const size_t wbytes = pad16(weights_.size() * sizeof(float)), bbytes = pad16(bands.size() * sizeof(int32_t)),
                   tbytes = pad16(tiles.size() * sizeof(float));
      weights_dev_.Reserve(wbytes + bbytes + tbytes + row_blocks.size() * sizeof(int32_t));
      char *base = static_cast<char *>(weights_dev_.data());
      bands_dev_ = reinterpret_cast<const int32_t *>(base + wbytes);
      tiles_dev_ = reinterpret_cast<const float *>(base + wbytes + bbytes);
      row_blocks_dev_ = reinterpret_cast<const int32_t *>(base + wbytes + bbytes + tbytes);
      KCHECK(daliamdMemcpyH2DAsync(base, weights_.data(), weights_.size() * sizeof(float), ws.stream));
      KCHECK(daliamdMemcpyH2DAsync(const_cast<int32_t *>(bands_dev_), bands.data(), bands.size() * sizeof(int32_t), ws.stream));
      KCHECK(daliamdMemcpyH2DAsync(const_cast<float *>(tiles_dev_), tiles.data(), tiles.size() * sizeof(float), ws.stream));
      KCHECK(daliamdMemcpyH2DAsync(const_cast<int32_t *>(row_blocks_dev_), row_blocks.data(), row_blocks.size() * sizeof(int32_t),
                                   ws.stream));
      KCHECK(daliamdStreamSynchronize(ws.stream));  // one-time upload from pageable memory, read by later iterations
    }
    if (def) {
      // fused: spectrogram + filter bank in one launch - here, or (deferred once more) by the ToDecibels behind
      daliamdSpecMelParams mp{};
      const bool use_valu = getenv("DALI_AMD_MEL_VALU") && atoi(getenv("DALI_AMD_MEL_VALU")) != 0;   // (benchmarks: the banded VALU variant)
      mp.mfma_tiles = use_valu ? nullptr : tiles_dev_;
      mp.row_blocks = row_blocks_dev_;
      mp.weights = static_cast<const float *>(weights_dev_.data());
      mp.bands = bands_dev_;
      mp.nfilter = nfilter_;
      mp.nbins = nbins_;
      if (defer_) {
        auto d = std::make_shared<DeferredAudio>(*def);
        d->has_mel = true;
        d->mel = mp;
        out.Resize({}, DALI_FLOAT);
        out.deferred_audio = d;
        return;
      }
      std::vector<daliamdSpectrogramDesc> descs = def->descs;
      for (int i = 0; i < n; i++) descs[i].out = static_cast<float *>(out.raw(i));
      auto *dev = static_cast<const daliamdSpectrogramDesc *>(uploader_.Upload(descs.data(), n * sizeof(descs[0]), ws.stream, ws.ring + 1));
      KCHECK(daliamdSpectrogramMelRun(ws.stream, dev, n, &def->params, def->window_dev, def->twiddles_dev, &mp, def->nwg));
      NoteLaunch(ws, use_valu ? "spectrogram_mel_fused" : "spectrogram_mel_fused_mfma");
      return;
    }
    for (int i = 0; i < n; i++) descs_[i].out = static_cast<float *>(out.raw(i));
    int nwg = 0;
    KCHECK(daliamdMelFilterBankSetup(descs_.data(), n, &nwg));
    auto *dev = static_cast<const daliamdMelDesc *>(uploader_.Upload(descs_.data(), n * sizeof(descs_[0]), ws.stream, ws.ring + 1));
    KCHECK(daliamdMelFilterBankRun(ws.stream, dev, n, nwg, static_cast<const float *>(weights_dev_.data()), bands_dev_, nfilter_,
                                   nbins_));
    NoteLaunch(ws, "mel_filter_bank_banded");
  }

 private:
  int nfilter_, nbins_ = 0, formula_ = 0;
  float sample_rate_, freq_low_, freq_high_;
  bool normalize_;
  std::vector<float> weights_, host_weights_;
  Buffer weights_dev_;
  const int32_t *bands_dev_ = nullptr, *row_blocks_dev_ = nullptr;
  const float *tiles_dev_ = nullptr;
  bool fused_input_ = false, defer_ = false;
  std::vector<daliamdMelDesc> descs_;
  DescUploader uploader_;
};
DALI_REGISTER_OPERATOR(MelFilterBank, MelFilterBankGpu, GPU);
DALI_REGISTER_OPERATOR(MelFilterBank, MelFilterBankGpu, CPU);

// =============================================================================================
// ToDecibels
// =============================================================================================
DALI_SCHEMA(ToDecibels)
    .DocStr("Converts a magnitude (real, positive) to the decibel scale: "
            "``min_ratio = pow(10, cutoff_db / multiplier); out[i] = multiplier * log10(max(min_ratio, input[i] / reference))``.")
    .NumInput(1)
    .NumOutput(1)
    .AddOptionalArg("multiplier", "Factor by which the logarithm is multiplied (10 for power, 20 for magnitude).", ArgValue::Float(10.0))
    .AddOptionalTypeArg("reference", "Reference magnitude. If not provided, the maximum value of the input is used.", ArgType::FLOAT)
    .AddOptionalArg("cutoff_db", "Minimum or cut-off ratio in dB.", ArgValue::Float(-200.0));

class ToDecibelsGpu : public OperatorBase {
 public:
  explicit ToDecibelsGpu(const OpSpec &spec) : OperatorBase(spec) {
    multiplier_ = (float)spec.GetFloat("multiplier");
    cutoff_ = (float)spec.GetFloat("cutoff_db");
    if (spec.TryArg("reference")) {
      reference_ = (float)spec.GetFloat("reference");
      DALI_ENFORCE(reference_ != 0, "`reference` argument can't be zero");
      DALI_ENFORCE(reference_ > 0, "`reference` must be positive");
    }
  }
  void ExpectFusedInput() { fused_input_ = true; }
  bool SetupImpl(std::vector<OutputDesc> &desc, const Workspace &ws) override {
    const TensorList &in = ws.Input(0);
    if (fused_input_ && ws.backend != OpType::CPU) {
      const DeferredAudio *def = in.deferred_audio.get();
      DALI_ENFORCE(def && def->has_mel, "internal: ToDecibels expected the deferred arguments of the MelFilterBank in front of it");
      desc[0].type = DALI_FLOAT;
      desc[0].shape.clear();
      for (auto &d : def->descs) desc[0].shape.push_back({def->mel.nfilter, d.num_windows});
      return true;
    }
    DALI_ENFORCE(in.type() == DALI_FLOAT, "ToDecibels expects float32 input");
    desc[0].type = DALI_FLOAT;
    desc[0].shape.clear();
    for (int i = 0; i < in.num_samples(); i++) desc[0].shape.push_back(in.shape(i));
    return true;
  }
  void RunImpl(Workspace &ws) override {
    const TensorList &in = ws.Input(0);
    TensorList &out = ws.Output(0);
    out.SetLayout(in.layout());
    if (const DeferredAudio *def = fused_input_ && ws.backend != OpType::CPU ? in.deferred_audio.get() : nullptr) {
      // spectrogram + filter bank + decibels.  With a given reference it is ONE launch; with the sample's maximum as the
      // reference the fused kernel leaves the mel energies in the output and their maxima in this operator's descriptor
      // table, and the element-wise pass runs in place behind it.
      out.SetLayout("ft");
      const int n = (int)def->descs.size();
      if (!n) return;
      std::vector<daliamdSpectrogramDesc> sd = def->descs;
      descs_.assign(n, daliamdDecibelDesc{});
      for (int i = 0; i < n; i++) {
        sd[i].out = static_cast<float *>(out.raw(i));
        descs_[i].in = descs_[i].out = sd[i].out;
        descs_[i].size = (int64_t)def->mel.nfilter * sd[i].num_windows;
      }
      daliamdSpecMelParams mp = def->mel;
      const bool by_max = !(reference_ > 0.0f);
      int nwg = 0;
      daliamdDecibelDesc *ddev = nullptr;
      if (by_max) {
        KCHECK(daliamdToDecibelsSetup(descs_.data(), n, &nwg));
        ddev = static_cast<daliamdDecibelDesc *>(uploader_.Upload(descs_.data(), n * sizeof(descs_[0]), ws.stream, ws.ring + 1));
        mp.max_bits = &ddev->max_bits;
        mp.max_stride = (int32_t)sizeof(daliamdDecibelDesc);
      } else {
        mp.decibels = 1;
        mp.multiplier = multiplier_; mp.reference = reference_; mp.cutoff_db = cutoff_;
      }
      auto *dev = static_cast<const daliamdSpectrogramDesc *>(spec_uploader_.Upload(sd.data(), n * sizeof(sd[0]), ws.stream, ws.ring + 1));
      KCHECK(daliamdSpectrogramMelRun(ws.stream, dev, n, &def->params, def->window_dev, def->twiddles_dev, &mp, def->nwg));
      NoteLaunch(ws, mp.mfma_tiles ? "spectrogram_mel_fused_mfma" : "spectrogram_mel_fused");
      if (by_max) {
        KCHECK(daliamdToDecibelsRun(ws.stream, ddev, n, nwg, multiplier_, -1.0f, cutoff_));
        NoteLaunch(ws, "to_decibels");
      }
      return;
    }
    int n = in.num_samples();
    if (!n) return;
    if (ws.backend == OpType::CPU) {
      for (int i = 0; i < n; i++) {
        const float *src = static_cast<const float *>(in.raw(i));
        float *dst = static_cast<float *>(out.raw(i));
        const int64_t size = volume(in.shape(i));
        ws.GetThreadPool().AddWork([this, src, dst, size](int) {
          if (daliamdToDecibelsHost(src, size, multiplier_, reference_, cutoff_, dst) != 0) DALI_FAIL(daliamdHostGetLastErrorMessage());
        }, size);
      }
      ws.GetThreadPool().RunAll();
      NoteLaunch(ws, "host_to_decibels");
      return;
    }
    descs_.assign(n, daliamdDecibelDesc{});
    for (int i = 0; i < n; i++) {
      descs_[i].in = static_cast<const float *>(in.raw(i));
      descs_[i].out = static_cast<float *>(out.raw(i));
      descs_[i].size = volume(in.shape(i));
    }
    int nwg = 0;
    KCHECK(daliamdToDecibelsSetup(descs_.data(), n, &nwg));
    auto *dev = static_cast<daliamdDecibelDesc *>(uploader_.Upload(descs_.data(), n * sizeof(descs_[0]), ws.stream, ws.ring + 1));
    KCHECK(daliamdToDecibelsRun(ws.stream, dev, n, nwg, multiplier_, reference_, cutoff_));
    NoteLaunch(ws, "to_decibels");
  }

 private:
  float multiplier_, cutoff_, reference_ = 0.0f;
  bool fused_input_ = false;
  std::vector<daliamdDecibelDesc> descs_;
  DescUploader uploader_, spec_uploader_;
};
DALI_REGISTER_OPERATOR(ToDecibels, ToDecibelsGpu, GPU);
DALI_REGISTER_OPERATOR(ToDecibels, ToDecibelsGpu, CPU);

void TryEnablePcm16Fusion(OperatorBase *decoder, OperatorBase *spectrogram) {
  if (getenv("DALI_AMD_NO_PCM16_FUSION") && atoi(getenv("DALI_AMD_NO_PCM16_FUSION")) != 0) return;
  auto *dec = dynamic_cast<AudioDecoderCpu *>(decoder);
  auto *spec = dynamic_cast<SpectrogramGpu *>(spectrogram);
  if (!dec || !spec || !dec->CanEmitPcm16()) return;
  dec->EmitPcm16();
  spec->AcceptPcm16();
}

bool TryEnableAudioFusion(OperatorBase *producer, OperatorBase *consumer) {
  if (getenv("DALI_AMD_NO_AUDIO_FUSION") && atoi(getenv("DALI_AMD_NO_AUDIO_FUSION")) != 0) return false;
  if (auto *spec = dynamic_cast<SpectrogramGpu *>(producer)) {
    auto *mel = dynamic_cast<MelFilterBankGpu *>(consumer);
    if (!mel || !spec->EnableFusion()) return false;
    mel->ExpectFusedInput();
    return true;
  }
  if (auto *mel = dynamic_cast<MelFilterBankGpu *>(producer)) {
    auto *db = dynamic_cast<ToDecibelsGpu *>(consumer);
    if (!db || !mel->HasFusedInput()) return false;
    mel->DeferToDecibels();
    db->ExpectFusedInput();
    return true;
  }
  return false;
}

// =============================================================================================
// MFCC (dali/operators/audio/mfcc/mfcc.cc:24-182): DCT along the frequency axis + liftering
// =============================================================================================
DALI_SCHEMA(MFCC)
    .DocStr("Computes Mel Frequency Cepstral Coefficients (MFCC) from a mel spectrogram.")
    .NumInput(1)
    .NumOutput(1)
    .AddOptionalArg("n_mfcc", "Number of MFCC coefficients.", ArgValue::Int(20))
    .AddOptionalArg("dct_type", "Discrete Cosine Transform type (1, 2, 3 or 4, as listed in "
                    "https://en.wikipedia.org/wiki/Discrete_cosine_transform#Formal_definition).", ArgValue::Int(2))
    .AddOptionalArg("normalize", "If set to True, the DCT uses an ortho-normal basis (not supported for dct_type=1).",
                    ArgValue::Bool(false))
    .AddOptionalArg("axis", "Axis over which the transform will be applied.", ArgValue::Int(0))
    .AddOptionalArg("lifter", "Cepstral filtering (liftering) coefficient: MFCC[i] *= 1 + sin(pi * (i + 1) / lifter) * lifter / 2; "
                    "0 disables it.", ArgValue::Float(0.0));

class MfccGpu : public OperatorBase {
 public:
  explicit MfccGpu(const OpSpec &spec) : OperatorBase(spec), tables_dev_(StorageDevice::GPU) {
    n_mfcc_ = (int)spec.GetInt("n_mfcc");
    dct_type_ = (int)spec.GetInt("dct_type");
    normalize_ = spec.GetBool("normalize");
    axis_ = (int)spec.GetInt("axis");
    lifter_ = (float)spec.GetFloat("lifter");
    DALI_ENFORCE(dct_type_ >= 1 && dct_type_ <= 4, "Unsupported DCT type: ", dct_type_, ". Supported types are: 1, 2, 3, 4");
    DALI_ENFORCE(!(normalize_ && dct_type_ == 1), "Ortho-normalization is not supported for DCT type I.");
  }
  bool SetupImpl(std::vector<OutputDesc> &desc, const Workspace &ws) override {
    const TensorList &in = ws.Input(0);
    DALI_ENFORCE(in.type() == DALI_FLOAT, "MFCC expects float32 input");
    int n = in.num_samples();
    descs_.clear();
    slices_.assign(n, {0, 0});
    desc[0].type = DALI_FLOAT;
    desc[0].shape.resize(n);
    // The sample is seen as [outer][n_in][inner] around `axis`: one descriptor per outer slice, the inner extent takes
    // the place of the frames (mfcc.cc:128-172, dct_cpu.cc:39-62: the transform runs along `axis`, every other extent is kept).
    for (int i = 0; i < n; i++) {
      const TensorShape &sh = in.shape(i);
      int ndim = (int)sh.size();
      DALI_ENFORCE(axis_ >= 0 && axis_ < ndim, "Axis ", axis_, " is out of bounds [0,", ndim, ")");
      int nin = (int)sh[axis_];
      DALI_ENFORCE(n_in_ == 0 || nin == n_in_, "All inputs must have the same number of mel bands (got ", nin, " after ", n_in_, ")");
      n_in_ = nin;
      int64_t outer = 1, inner = 1;
      for (int k = 0; k < axis_; k++) outer *= sh[k];
      for (int k = axis_ + 1; k < ndim; k++) inner *= sh[k];
      DALI_ENFORCE(inner < (1ll << 31) && outer < (1ll << 24), "MFCC: the input is too large");
      slices_[i] = {outer, inner};
    }
    ndct_ = n_mfcc_ <= 0 || n_mfcc_ > n_in_ ? n_in_ : n_mfcc_;  // dct_cpu.cc:56-58
    for (int i = 0; i < n; i++) {
      desc[0].shape[i] = in.shape(i);
      desc[0].shape[i][axis_] = ndct_;
      const float *base = static_cast<const float *>(in.raw(i));
      for (int64_t o = 0; o < slices_[i].first; o++) {
        daliamdMelDesc d{};
        d.in = base + o * n_in_ * slices_[i].second;
        d.frames = (int)slices_[i].second;
        descs_.push_back(d);
      }
    }
    return true;
  }
  void RunImpl(Workspace &ws) override {
    TensorList &out = ws.Output(0);
    out.SetLayout(ws.Input(0).layout());
    int n = (int)descs_.size();
    if (!n) return;
    if (ws.backend == OpType::CPU) {
      if (host_tables_n_in_ != n_in_) {
        host_tables_.resize((size_t)ndct_ * n_in_ + ndct_);
        KCHECK(daliamdDctTable(dct_type_, normalize_, n_in_, ndct_, host_tables_.data()));
        daliamdLifterCoeffs(lifter_, ndct_, host_tables_.data() + (size_t)ndct_ * n_in_);
        host_tables_n_in_ = n_in_;
      }
      for (size_t i = 0, k = 0; i < slices_.size(); i++) {
        float *base = static_cast<float *>(out.raw((int)i));
        for (int64_t o = 0; o < slices_[i].first; o++, k++) {
          const float *src = descs_[k].in;
          float *dst = base + o * ndct_ * slices_[i].second;
          const int64_t inner = slices_[i].second;
          ws.GetThreadPool().AddWork([this, src, dst, inner](int) {
            const float *table = host_tables_.data();
            if (daliamdDctHost(src, n_in_, inner, table, lifter_ != 0.0f ? table + (size_t)ndct_ * n_in_ : nullptr, ndct_, dst) != 0)
              DALI_FAIL(daliamdHostGetLastErrorMessage());
          }, inner);
        }
      }
      ws.GetThreadPool().RunAll();
      NoteLaunch(ws, "host_mfcc_dct");
      return;
    }
    if (tables_n_in_ != n_in_) {
      std::vector<float> host((size_t)ndct_ * n_in_ + ndct_);
      KCHECK(daliamdDctTable(dct_type_, normalize_, n_in_, ndct_, host.data()));
      daliamdLifterCoeffs(lifter_, ndct_, host.data() + (size_t)ndct_ * n_in_);
      tables_dev_.Reserve(host.size() * sizeof(float));
      KCHECK(daliamdMemcpyH2DAsync(tables_dev_.data(), host.data(), host.size() * sizeof(float), ws.stream));
      KCHECK(daliamdStreamSynchronize(ws.stream));  // one-time upload from pageable memory, read by later iterations
      tables_n_in_ = n_in_;
    }
    for (size_t i = 0, k = 0; i < slices_.size(); i++) {
      float *base = static_cast<float *>(out.raw((int)i));
      for (int64_t o = 0; o < slices_[i].first; o++) descs_[k++].out = base + o * ndct_ * slices_[i].second;
    }
    int nwg = 0;
    KCHECK(daliamdMelFilterBankSetup(descs_.data(), n, &nwg));
    auto *dev = static_cast<const daliamdMelDesc *>(uploader_.Upload(descs_.data(), n * sizeof(descs_[0]), ws.stream, ws.ring + 1));
    const float *table = static_cast<const float *>(tables_dev_.data());
    KCHECK(daliamdDctRun(ws.stream, dev, n, nwg, table, lifter_ != 0.0f ? table + (size_t)ndct_ * n_in_ : nullptr, ndct_, n_in_));
    NoteLaunch(ws, "mfcc_dct");
  }

 private:
  int n_mfcc_, dct_type_, axis_, n_in_ = 0, ndct_ = 0, tables_n_in_ = -1, host_tables_n_in_ = -1;
  std::vector<float> host_tables_;
  bool normalize_;
  float lifter_;
  Buffer tables_dev_;
  std::vector<daliamdMelDesc> descs_;
  std::vector<std::pair<int64_t, int64_t>> slices_;   // per sample: outer slices, inner extent
  DescUploader uploader_;
};
DALI_REGISTER_OPERATOR(MFCC, MfccGpu, GPU);
DALI_REGISTER_OPERATOR(MFCC, MfccGpu, CPU);

// =============================================================================================
// AudioResample (dali/operators/audio/resample.cc:24-140, resample.h:30-140)
// =============================================================================================
DALI_SCHEMA(AudioResample)
    .DocStr("Resamples an audio signal.\n\nThe resampling is achieved by applying a sinc filter with Hann window with an extent "
            "controlled by the `quality` argument. The resampling ratio can be specified directly or as a ratio of target to "
            "source sampling rate, or calculated from the ratio of the requested output length to the input length.")
    .NumInput(1)
    .NumOutput(1)
    .AddOptionalTypeArg("in_rate", "Input sampling rate (only the ratio to `out_rate` matters).", ArgType::FLOAT, true)
    .AddOptionalTypeArg("out_rate", "Output sampling rate.", ArgType::FLOAT, true)
    .AddOptionalTypeArg("scale", "The scaling factor: the ratio of the target sampling rate to the source sampling rate.",
                        ArgType::FLOAT, true)
    .AddOptionalTypeArg("out_length", "The requested output length, in samples.", ArgType::INT, true)
    .AddOptionalArg("quality", "Resampling quality, where 0 is the lowest, and 100 is the highest: 0 gives 3 lobes of the sinc "
                    "filter, 50 gives 16 lobes, and 100 gives 64 lobes.", ArgValue::Float(50.0))
    .AddOptionalTypeArg("dtype", "The output type. If not specified, the output type is the same as the input type. Integer "
                        "samples are normalised: signed types to -1..1, unsigned types to 0..1.", ArgType::INT);
DALI_SCHEMA(experimental__AudioResample).DocStr("Legacy alias for :meth:`audio_resample`.").AddParent("AudioResample").NumInput(1).NumOutput(1);

class AudioResampleGpu : public OperatorBase {
 public:
  explicit AudioResampleGpu(const OpSpec &spec) : OperatorBase(spec), lookup_dev_(StorageDevice::GPU) {
    auto given = [&](const char *name) { return spec.Args().count(name) != 0 || spec.HasTensorArgument(name); };
    has_rates_ = given("in_rate") || given("out_rate");
    has_scale_ = given("scale");
    has_len_ = given("out_length");
    DALI_ENFORCE(given("in_rate") == given("out_rate"),
                 "The parameters ``in_rate`` and ``out_rate`` must be specified together.");
    DALI_ENFORCE((int)has_rates_ + (int)has_scale_ + (int)has_len_ <= 1,
                 "The sampling rates, ``scale`` and ``out_length`` cannot be used together.");
    DALI_ENFORCE(has_rates_ || has_scale_ || has_len_,
                 "No resampling factor specified! Please supply either the scale, the output length or the input and output "
                 "sampling rates.");
    quality_ = (float)spec.GetFloat("quality");
    DALI_ENFORCE(quality_ >= 0 && quality_ <= 100, "``quality`` out of range: ", quality_, "\nValid range is [0..100].");
    if (const ArgValue *d = spec.TryArg("dtype")) {
      dtype_ = (DALIDataType)d->i;
      DALI_ENFORCE(KernelType(dtype_) >= 0, "Unsupported output type: ", (int)dtype_,
                   "\nSupported types are : int8, uint8, int16, uint16, int32, uint32, float");
    }
  }
  // daliamdDType_t of an audio sample type (AUDIO_RESAMPLE_TYPES, resample.h:28), -1 for anything else
  static int KernelType(DALIDataType t) {
    switch (t) {
      case DALI_INT8: return DALIAMD_INT8;
      case DALI_UINT8: return DALIAMD_UINT8;
      case DALI_INT16: return DALIAMD_INT16;
      case DALI_UINT16: return DALIAMD_UINT16;
      case DALI_INT32: return DALIAMD_INT32;
      case DALI_UINT32: return DALIAMD_UINT32;
      case DALI_FLOAT: return DALIAMD_FLOAT;
      default: return -1;
    }
  }
  static bool IsUnsignedType(DALIDataType t) { return t == DALI_UINT8 || t == DALI_UINT16 || t == DALI_UINT32; }
  bool SetupImpl(std::vector<OutputDesc> &desc, const Workspace &ws) override {
    const TensorList &in = ws.Input(0);
    in_type_ = in.type();
    DALI_ENFORCE(KernelType(in_type_) >= 0, "Unsupported input type: ", (int)in_type_,
                 "\nSupported types are : int8, uint8, int16, uint16, int32, uint32, float");
    out_type_ = dtype_ == DALI_NO_TYPE ? in_type_ : dtype_;
    // ConvertInput (resample.cc:160-192): what happens to the samples on their way to floats
    const bool out_unsigned = IsUnsignedType(out_type_), in_unsigned = IsUnsignedType(in_type_);
    if (in_type_ == DALI_FLOAT) in_mode_ = out_unsigned ? 1 : -1;                     // -1: the input is used as it is
    else in_mode_ = out_unsigned && !in_unsigned ? 1 : !out_unsigned && in_unsigned ? 2 : 0;
    int n = in.num_samples();
    std::vector<float> in_rate, out_rate, scale;
    std::vector<int> out_len;
    if (has_rates_) {
      in_rate = GetPerSampleFloat(spec_, ws, "in_rate", n);
      out_rate = GetPerSampleFloat(spec_, ws, "out_rate", n);
    } else if (has_scale_) {
      scale = GetPerSampleFloat(spec_, ws, "scale", n);
    } else {
      out_len = GetPerSampleInt(spec_, ws, "out_length", n);
    }
    descs_.assign(n, daliamdAudioResampleDesc{});
    in_raw_.assign(n, nullptr);
    desc[0].type = out_type_;
    desc[0].shape.resize(n);
    for (int i = 0; i < n; i++) {
      const TensorShape &s = in.shape(i);
      DALI_ENFORCE(s.size() == 1 || s.size() == 2,
                   "Audio resampling supports only time series data, with an optional innermost channel dimension.");
      auto &d = descs_[i];
      in_raw_[i] = in.raw(i);
      d.in = static_cast<const float *>(in.raw(i));   // replaced by the float copy when the samples are converted
      d.in_length = s[0];
      d.channels = s.size() == 2 ? (int)s[1] : 1;
      if (has_rates_) {
        DALI_ENFORCE(in_rate[i] > 0, "Input sampling rates must be positive. Got in_rate == ", in_rate[i]);
        DALI_ENFORCE(out_rate[i] >= 0, "Output sampling rates must be non-negative. Got out_rate == ", out_rate[i]);
        d.in_rate = in_rate[i];
        d.out_rate = out_rate[i];
        d.out_length = (int64_t)std::ceil(d.in_length * d.out_rate / d.in_rate);  // resampled_length, resampling.h:101-103
      } else if (has_scale_) {
        DALI_ENFORCE(scale[i] >= 0, "The scaling factor must be non-negative. Got scale == ", scale[i]);
        d.in_rate = 1.0;
        d.out_rate = scale[i];
        d.out_length = (int64_t)std::ceil(d.in_length * d.out_rate / d.in_rate);
      } else {
        DALI_ENFORCE(!(d.in_length == 0 && out_len[i] != 0), "Cannot produce a non-empty signal from an empty input.\nError at sample ", i);
        d.in_rate = d.in_length ? (double)d.in_length : 1.0;
        d.out_rate = out_len[i] ? (double)out_len[i] : 1.0;
        d.out_length = out_len[i];
      }
      if (d.out_length == 0) d.out_rate = 1.0;  // nothing to produce; keeps the kernel-side validation simple
      desc[0].shape[i] = s.size() == 2 ? TensorShape{d.out_length, s[1]} : TensorShape{d.out_length};
    }
    return true;
  }
  void RunImpl(Workspace &ws) override {
    TensorList &out = ws.Output(0);
    out.SetLayout(ws.Input(0).layout());
    int n = (int)descs_.size();
    if (!n) return;
    const int kin = KernelType(in_type_), kout = KernelType(out_type_);
    const bool cvt_in = in_mode_ >= 0, cvt_out = out_type_ != DALI_FLOAT;
    if (ws.backend == OpType::CPU) {  // one thread-pool task per sample on the host kernels
      for (int i = 0; i < n; i++) {
        void *out_raw = out.raw(i);
        ws.GetThreadPool().AddWork([this, i, out_raw, kin, kout, cvt_in, cvt_out](int) {
          const auto &d = descs_[i];
          std::vector<float> fin, fout;
          const float *src = static_cast<const float *>(in_raw_[i]);
          if (cvt_in) {
            fin.resize((size_t)d.in_length * d.channels);
            if (daliamdConvertNormHost(in_raw_[i], kin, fin.data(), DALIAMD_FLOAT, (int64_t)fin.size(), in_mode_) != 0)
              DALI_FAIL(daliamdHostGetLastErrorMessage());
            src = fin.data();
          }
          float *dst = static_cast<float *>(out_raw);
          if (cvt_out) {
            fout.resize((size_t)d.out_length * d.channels);
            dst = fout.data();
          }
          if (daliamdAudioResampleHost(src, d.in_length, d.channels, d.in_rate, d.out_rate, quality_, dst, d.out_length) != 0)
            DALI_FAIL(daliamdHostGetLastErrorMessage());
          if (cvt_out && daliamdConvertNormHost(fout.data(), DALIAMD_FLOAT, out_raw, kout, (int64_t)fout.size(), 0) != 0)
            DALI_FAIL(daliamdHostGetLastErrorMessage());
        }, descs_[i].out_length);
      }
      ws.GetThreadPool().RunAll();
      NoteLaunch(ws, "host_audio_resample");
      return;
    }
    if (!lookup_size_) {
      lobes_ = daliamdAudioResampleLobes(quality_);
      std::vector<float> lookup((size_t)lobes_ * 64 + 1 + 5);
      KCHECK(daliamdAudioResampleWindow(lobes_, lookup.data(), (int)lookup.size(), &lookup_size_, &wscale_, &wcenter_));
      lookup_dev_.Reserve(lookup.size() * sizeof(float));
      KCHECK(daliamdMemcpyH2DAsync(lookup_dev_.data(), lookup.data(), lookup.size() * sizeof(float), ws.stream));
      KCHECK(daliamdStreamSynchronize(ws.stream));  // one-time upload from pageable memory
    }
    // float copies of typed inputs / float results of typed outputs: scratch behind a table slot, 16-byte aligned per sample
    size_t in_elems = 0, out_elems = 0;
    std::vector<size_t> in_off(n), out_off(n);
    for (int i = 0; i < n; i++) {
      in_off[i] = in_elems;
      out_off[i] = out_elems;
      in_elems += ((size_t)descs_[i].in_length * descs_[i].channels + 3) & ~(size_t)3;
      out_elems += ((size_t)descs_[i].out_length * descs_[i].channels + 3) & ~(size_t)3;
    }
    float *fin = nullptr, *fout = nullptr;
    if (cvt_in || cvt_out) {  // the slot of a (dummy) table upload: reused only when the iteration `ring` steps back is done
      const uint64_t tag = 0;
      const size_t in_bytes = cvt_in ? (in_elems * sizeof(float) + 255) & ~(size_t)255 : 0;
      scratch_up_.Upload(&tag, sizeof(tag), ws.stream, ws.ring + 1, in_bytes + std::max<size_t>(out_elems, 4) * sizeof(float));
      char *base = static_cast<char *>(scratch_up_.Scratch());
      fin = reinterpret_cast<float *>(base);
      fout = reinterpret_cast<float *>(base + in_bytes);
    }
    if (cvt_in) {
      cvt_.assign(n, daliamdConvertNormDesc{});
      for (int i = 0; i < n; i++) {
        cvt_[i].in = in_raw_[i];
        cvt_[i].out = fin + in_off[i];
        cvt_[i].count = descs_[i].in_length * descs_[i].channels;
        descs_[i].in = fin + in_off[i];
      }
      int cwg = 0;
      KCHECK(daliamdConvertNormSetup(cvt_.data(), n, &cwg));
      auto *cdev = static_cast<const daliamdConvertNormDesc *>(cvt_uploader_[0].Upload(cvt_.data(), n * sizeof(cvt_[0]), ws.stream, ws.ring + 1));
      KCHECK(daliamdConvertNormRun(ws.stream, cdev, n, cwg, kin, DALIAMD_FLOAT, in_mode_));
      NoteLaunch(ws, "audio_samples_to_float");
    }
    for (int i = 0; i < n; i++) descs_[i].out = cvt_out ? fout + out_off[i] : static_cast<float *>(out.raw(i));
    int nwg = 0;
    KCHECK(daliamdAudioResampleSetup(descs_.data(), n, &nwg));
    auto *dev = static_cast<const daliamdAudioResampleDesc *>(uploader_.Upload(descs_.data(), n * sizeof(descs_[0]), ws.stream, ws.ring + 1));
    KCHECK(daliamdAudioResampleRun(ws.stream, dev, n, nwg, static_cast<const float *>(lookup_dev_.data()), lookup_size_, wscale_,
                                   wcenter_, lobes_));
    if (cvt_out) {
      NoteLaunch(ws, "audio_resample");
      cvt_.assign(n, daliamdConvertNormDesc{});
      for (int i = 0; i < n; i++) {
        cvt_[i].in = fout + out_off[i];
        cvt_[i].out = out.raw(i);
        cvt_[i].count = descs_[i].out_length * descs_[i].channels;
      }
      int cwg = 0;
      KCHECK(daliamdConvertNormSetup(cvt_.data(), n, &cwg));
      auto *cdev = static_cast<const daliamdConvertNormDesc *>(cvt_uploader_[1].Upload(cvt_.data(), n * sizeof(cvt_[0]), ws.stream, ws.ring + 1));
      KCHECK(daliamdConvertNormRun(ws.stream, cdev, n, cwg, DALIAMD_FLOAT, kout, 0));
      NoteLaunch(ws, "audio_samples_from_float");
      return;
    }
    NoteLaunch(ws, "audio_resample");
  }

 private:
  bool has_rates_, has_scale_, has_len_;
  float quality_, wscale_ = 0, wcenter_ = 0;
  int lobes_ = 0, lookup_size_ = 0, in_mode_ = -1;
  DALIDataType dtype_ = DALI_NO_TYPE, in_type_ = DALI_FLOAT, out_type_ = DALI_FLOAT;
  Buffer lookup_dev_;
  std::vector<daliamdAudioResampleDesc> descs_;
  std::vector<daliamdConvertNormDesc> cvt_;
  std::vector<const void *> in_raw_;
  DescUploader uploader_, cvt_uploader_[2], scratch_up_;
};
DALI_REGISTER_OPERATOR(AudioResample, AudioResampleGpu, GPU);
DALI_REGISTER_OPERATOR(experimental__AudioResample, AudioResampleGpu, GPU);
DALI_REGISTER_OPERATOR(AudioResample, AudioResampleGpu, CPU);  // same class: the host kernel when run on the CPU
DALI_REGISTER_OPERATOR(experimental__AudioResample, AudioResampleGpu, CPU);

}  // namespace daliamd_host
