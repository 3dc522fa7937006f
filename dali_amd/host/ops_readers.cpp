// Readers for the container formats ImageNet-scale data sets are shipped in, feeding the same decoders as readers.file
// and sharing its Loader (shards, shuffle reservoir, last-batch padding, checkpoints - ops.h):
//   readers.tfrecord    dali/operators/reader/tfrecord_reader_op.cc:1-166, parser/tfrecord_parser.h:40-196,
//                       loader/indexed_file_loader.h (index file: one "offset size" line per record)
//   readers.mxnet       dali/operators/reader/mxnet_reader_op.cc, loader/recordio_loader.h:1-177,
//                       parser/recordio_parser.h:31-186 (RecordIO + ImageRecordIOHeader, multi-segment records)
//   readers.webdataset  dali/operators/reader/webdataset_reader_op.cc:22-188, loader/webdataset_loader.cc:1-546
//                       (tar archives, samples = files sharing a base name, optional index v1.1 / v1.2)
// Records are read with pread() into the operator's output on the thread pool; the files stay open.
#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <cstring>
#include <fstream>
#include <map>

#include "ops.h"

namespace daliamd_host {
namespace {

struct OpenFiles {
  std::vector<std::string> paths;
  std::vector<int> fds;
  std::vector<int64_t> sizes;
  explicit OpenFiles(const std::vector<std::string> &p) : paths(p) {
    for (auto &path : paths) {
      int fd = open(path.c_str(), O_RDONLY);
      DALI_ENFORCE(fd >= 0, "Could not open file ", path);
      struct stat st;
      DALI_ENFORCE(fstat(fd, &st) == 0, "Could not stat file ", path);
      fds.push_back(fd);
      sizes.push_back((int64_t)st.st_size);
    }
  }
  ~OpenFiles() { for (int fd : fds) close(fd); }
  void Read(int file, int64_t offset, void *dst, int64_t size) const {
    int64_t got = 0;
    while (got < size) {
      ssize_t r = pread(fds[file], static_cast<char *>(dst) + got, (size_t)(size - got), (off_t)(offset + got));
      if (r <= 0) break;
      got += r;
    }
    DALI_ENFORCE(got == size, "Failed to read ", size, " bytes at offset ", offset, " of ", paths[file]);
  }
};

std::vector<std::string> StrVec(const OpSpec &spec, const std::string &name) {
  const ArgValue *a = spec.TryArg(name);
  if (!a) return {};
  return a->type == ArgType::STRING ? std::vector<std::string>{a->s} : a->sv;
}

// ---- protobuf wire format, just enough for tensorflow.Example (tensorflow/core/example/{example,feature}.proto) ----
struct PbView {
  const uint8_t *p, *end;
  bool ok = true;
  uint64_t Varint() {
    uint64_t v = 0;
    for (int shift = 0; shift < 64 && p < end; shift += 7) {
      const uint8_t b = *p++;
      v |= (uint64_t)(b & 127) << shift;
      if (!(b & 128)) return v;
    }
    ok = false;
    return 0;
  }
  PbView Bytes() {
    const uint64_t n = Varint();
    if (!ok || n > (uint64_t)(end - p)) { ok = false; return PbView{end, end, false}; }
    PbView v{p, p + n, true};
    p += n;
    return v;
  }
  void Skip(int wire) {
    if (wire == 0) Varint();
    else if (wire == 1) { if (end - p < 8) ok = false; else p += 8; }
    else if (wire == 2) Bytes();
    else if (wire == 5) { if (end - p < 4) ok = false; else p += 4; }
    else ok = false;
  }
  bool More() const { return ok && p < end; }
};

struct TfFeature {  // one Feature message: exactly one of the three lists
  std::vector<PbView> bytes;
  std::vector<float> floats;
  std::vector<int64_t> ints;
};
bool ParseFeature(PbView f, TfFeature *out) {
  while (f.More()) {
    const uint64_t key = f.Varint();
    const int field = (int)(key >> 3), wire = (int)(key & 7);
    if (wire != 2 || field < 1 || field > 3) { f.Skip(wire); continue; }
    PbView list = f.Bytes();  // BytesList / FloatList / Int64List, each with `repeated value = 1`
    while (list.More()) {
      const uint64_t k = list.Varint();
      const int lf = (int)(k >> 3), lw = (int)(k & 7);
      if (lf != 1) { list.Skip(lw); continue; }
      if (field == 1 && lw == 2) {
        out->bytes.push_back(list.Bytes());
      } else if (field == 2 && lw == 2) {  // packed floats
        PbView pk = list.Bytes();
        for (; pk.end - pk.p >= 4; pk.p += 4) { float v; memcpy(&v, pk.p, 4); out->floats.push_back(v); }
      } else if (field == 2 && lw == 5) {
        if (list.end - list.p < 4) return false;
        float v; memcpy(&v, list.p, 4); list.p += 4; out->floats.push_back(v);
      } else if (field == 3 && lw == 2) {  // packed varints
        PbView pk = list.Bytes();
        while (pk.More()) out->ints.push_back((int64_t)pk.Varint());
        if (!pk.ok) return false;
      } else if (field == 3 && lw == 0) {
        out->ints.push_back((int64_t)list.Varint());
      } else {
        list.Skip(lw);
      }
    }
    if (!list.ok) return false;
  }
  return f.ok;
}
// Example { Features features = 1 }, Features { map<string, Feature> feature = 1 } -> name -> Feature view
bool ParseExample(PbView ex, std::map<std::string, PbView> *features) {
  while (ex.More()) {
    const uint64_t key = ex.Varint();
    if ((key >> 3) != 1 || (key & 7) != 2) { ex.Skip((int)(key & 7)); continue; }
    PbView fs = ex.Bytes();
    while (fs.More()) {
      const uint64_t k = fs.Varint();
      if ((k >> 3) != 1 || (k & 7) != 2) { fs.Skip((int)(k & 7)); continue; }
      PbView entry = fs.Bytes();  // map entry: key = 1 (string), value = 2 (Feature)
      std::string name;
      PbView value{nullptr, nullptr, true};
      while (entry.More()) {
        const uint64_t ek = entry.Varint();
        if ((ek & 7) != 2) { entry.Skip((int)(ek & 7)); continue; }
        PbView v = entry.Bytes();
        if ((ek >> 3) == 1) name.assign(reinterpret_cast<const char *>(v.p), (size_t)(v.end - v.p));
        else if ((ek >> 3) == 2) value = v;
      }
      if (!entry.ok) return false;
      (*features)[name] = value;
    }
    if (!fs.ok) return false;
  }
  return ex.ok;
}

struct Record { int64_t offset, size; int file; };

}  // namespace

// =============================================================================================
// readers.tfrecord
// =============================================================================================
DALI_SCHEMA(readers__TFRecord)
    .DocStr("Reads samples from a TensorFlow TFRecord file.  `features` (Python: a dictionary name -> "
            "dali_amd.tfrecord.FixedLenFeature / VarLenFeature) is passed as `feature_names` plus per-feature kind, "
            "type and shape vectors; one output per feature, in that order.  A feature that is missing in a record "
            "gives an empty tensor.")
    .NumInput(0)
    .NumOutput(1)
    .AddArg("path", "List of paths to TFRecord files.", ArgType::STRING_VEC)
    .AddArg("index_path", "List of paths to index files (one per TFRecord file; lines `offset size`, as written by "
            "tfrecord2idx).", ArgType::STRING_VEC)
    .AddArg("feature_names", "Names of the features to extract.", ArgType::STRING_VEC)
    .AddArg("feature_dtypes", "Per feature: INT64, FLOAT or UINT8 (= string).", ArgType::INT_VEC)
    .AddArg("feature_has_shape", "Per feature: 1 = FixedLenFeature (fixed shape), 0 = VarLenFeature.", ArgType::INT_VEC)
    .AddArg("feature_ndims", "Per feature: number of entries it has in `feature_shapes` (the fixed shape, or the partial "
            "shape of a VarLenFeature).", ArgType::INT_VEC)
    .AddOptionalTypeArg("feature_shapes", "All shapes, concatenated.", ArgType::INT_VEC)
    .AddOptionalArg("use_o_direct", "Ignored.", ArgValue::Bool(false))
    .AddOptionalArg("shuffle_after_epoch", "Not supported (must be False).", ArgValue::Bool(false))
    .AddParent("LoaderBase");
DALI_SCHEMA(TFRecordReader).DocStr("Legacy alias of readers.tfrecord").NumInput(0).NumOutput(1).AddParent("readers__TFRecord");

class TFRecordReaderOp : public OperatorBase {
 public:
  explicit TFRecordReaderOp(const OpSpec &spec)
      : OperatorBase(spec), loader_(spec), files_(StrVec(spec, "path")), names_(StrVec(spec, "feature_names")) {
    DALI_ENFORCE(!spec.GetBool("shuffle_after_epoch"), "readers.tfrecord: shuffle_after_epoch is not supported");
    const auto index_paths = StrVec(spec, "index_path");
    DALI_ENFORCE(index_paths.size() == files_.paths.size(), "Number of index files needs to match the number of data files");
    auto dt = spec.GetIntVec("feature_dtypes"), hs = spec.GetIntVec("feature_has_shape"), nd = spec.GetIntVec("feature_ndims");
    std::vector<int64_t> shapes;
    if (spec.TryArg("feature_shapes")) shapes = spec.GetIntVec("feature_shapes");
    DALI_ENFORCE(!names_.empty(), "No features provided");
    DALI_ENFORCE(dt.size() == names_.size() && hs.size() == names_.size() && nd.size() == names_.size(),
                 "Number of features needs to match number of feature names.");
    size_t pos = 0;
    for (size_t i = 0; i < names_.size(); i++) {
      FeatureSpec f;
      f.type = (DALIDataType)dt[i];
      DALI_ENFORCE(f.type == DALI_INT64 || f.type == DALI_FLOAT || f.type == DALI_UINT8,
                   "readers.tfrecord: feature \"", names_[i], "\": unsupported type ", TypeName(f.type));
      f.has_shape = hs[i] != 0;
      DALI_ENFORCE(nd[i] >= 0 && (size_t)nd[i] <= shapes.size() - pos, "readers.tfrecord: inconsistent feature shapes");
      f.shape.assign(shapes.begin() + pos, shapes.begin() + pos + nd[i]);
      pos += (size_t)nd[i];
      features_.push_back(f);
    }
    for (size_t k = 0; k < index_paths.size(); k++) {
      std::ifstream idx(index_paths[k]);
      DALI_ENFORCE(idx.good(), "Could not open index file ", index_paths[k]);
      int64_t off, size;
      while (idx >> off >> size) {
        DALI_ENFORCE(off >= 0 && size >= 0 && off <= files_.sizes[k] && size <= files_.sizes[k] - off, "Index file ", index_paths[k],
                     " does not describe ", files_.paths[k], " (record at ", off, " + ", size, ")");
        records_.push_back({off, size, (int)k});
      }
    }
    DALI_ENFORCE(!records_.empty(), "Content of index files should not be empty");
    loader_.Init((int64_t)records_.size());
  }
  ReaderMeta GetReaderMeta() const override { return loader_.Meta(); }
  std::string SaveState() const override { return loader_.Save(); }
  void RestoreState(const std::string &s) override { loader_.Restore(s); }
  bool SetupImpl(std::vector<OutputDesc> &, const Workspace &) override { return false; }

  void RunImpl(Workspace &ws) override {
    const int n = max_batch_size_, nf = (int)features_.size();
    DALI_ENFORCE((int)ws.outputs.size() == nf, "readers.tfrecord: ", nf, " features but ", ws.outputs.size(), " outputs");
    std::vector<int64_t> picks(n);
    for (int i = 0; i < n; i++) picks[i] = loader_.NextIndex(i == 0);
    // pass 1 (thread pool): read + parse every record, note where each feature's data lies
    raw_.resize(n);
    parsed_.assign((size_t)n * nf, Parsed{});
    for (int i = 0; i < n; i++) {
      ws.GetThreadPool().AddWork([this, &picks, i, nf](int) {
        const Record &r = records_[picks[i]];
        auto &buf = raw_[i];
        buf.resize((size_t)r.size);
        files_.Read(r.file, r.offset, buf.data(), r.size);
        const std::string where = make_string(files_.paths[r.file], " at index ", r.offset);
        DALI_ENFORCE(r.size >= 16, "Error while parsing TFRecord file: ", where, " (record is too short: ", r.size,
                     " bytes, minimum is 16 bytes).");
        uint64_t length;
        memcpy(&length, buf.data(), 8);
        DALI_ENFORCE(length <= (uint64_t)(r.size - 16), "Error while parsing TFRecord file: ", where,
                     " (record payload length: ", length, " bytes, available payload: ", r.size - 16, " bytes).");
        std::map<std::string, PbView> feats;
        DALI_ENFORCE(ParseExample(PbView{buf.data() + 12, buf.data() + 12 + length, true}, &feats),
                     "Error while parsing TFRecord file: ", where, " (raw data length: ", length, " bytes).");
        for (int f = 0; f < nf; f++) {
          Parsed &p = parsed_[(size_t)i * nf + f];
          auto it = feats.find(names_[f]);
          if (it == feats.end()) continue;  // empty tensor
          p.present = true;
          DALI_ENFORCE(ParseFeature(it->second, &p.value), "Error while parsing TFRecord file: ", where, " (feature \"",
                       names_[f], "\")");
        }
      }, records_[picks[i]].size);
    }
    ws.GetThreadPool().RunAll();
    // shapes, allocation, copy
    for (int f = 0; f < nf; f++) {
      const FeatureSpec &fs = features_[f];
      std::vector<TensorShape> shapes(n);
      for (int i = 0; i < n; i++) {
        const Parsed &p = parsed_[(size_t)i * nf + f];
        if (!p.present) { shapes[i] = TensorShape{0}; continue; }
        if (fs.type == DALI_UINT8) {
          int64_t vol = 1;
          for (int64_t e : fs.shape) vol *= e;
          DALI_ENFORCE(fs.has_shape && vol <= 1, "Tensors of strings are not supported.");
          DALI_ENFORCE(!p.value.bytes.empty(), "Feature \"", names_[f], "\" holds no bytes value");
          shapes[i] = TensorShape{(int64_t)(p.value.bytes[0].end - p.value.bytes[0].p)};
          continue;
        }
        const int64_t count = fs.type == DALI_INT64 ? (int64_t)p.value.ints.size() : (int64_t)p.value.floats.size();
        if (fs.has_shape) {
          shapes[i] = TensorShape(fs.shape.begin(), fs.shape.end());
          DALI_ENFORCE(count <= volume(shapes[i]), "Output tensor shape is too small. Expected at least ", count, " elements.");
        } else if (!fs.shape.empty()) {  // partial shape: the outermost extent is inferred
          int64_t m = 1;
          for (int64_t e : fs.shape) m *= e;
          DALI_ENFORCE(m > 0 && count % m == 0, "Feature size not matching partial shape");
          shapes[i] = TensorShape{count / m};
          shapes[i].insert(shapes[i].end(), fs.shape.begin(), fs.shape.end());
        } else {
          shapes[i] = TensorShape{count};
        }
      }
      TensorList &out = ws.Output(f);
      out.Resize(shapes, fs.type);
      out.source_info.resize(n);
      for (int i = 0; i < n; i++) {
        const Record &r = records_[picks[i]];
        out.source_info[i] = make_string(files_.paths[r.file], " at index ", r.offset);
        const Parsed &p = parsed_[(size_t)i * nf + f];
        if (!p.present) continue;
        // (an empty feature has no storage: memcpy must not see its null pointer, even with a length of zero)
        if (fs.type == DALI_UINT8) {
          if (out.nbytes(i)) memcpy(out.raw(i), p.value.bytes[0].p, (size_t)out.nbytes(i));
        } else if (fs.type == DALI_INT64) {
          memset(out.raw(i), 0, (size_t)out.nbytes(i));
          if (!p.value.ints.empty()) memcpy(out.raw(i), p.value.ints.data(), p.value.ints.size() * sizeof(int64_t));
        } else {
          memset(out.raw(i), 0, (size_t)out.nbytes(i));
          if (!p.value.floats.empty()) memcpy(out.raw(i), p.value.floats.data(), p.value.floats.size() * sizeof(float));
        }
      }
    }
  }

 private:
  struct FeatureSpec { DALIDataType type; bool has_shape; std::vector<int64_t> shape; };
  struct Parsed { bool present = false; TfFeature value; };
  Loader loader_;
  OpenFiles files_;
  std::vector<std::string> names_;
  std::vector<FeatureSpec> features_;
  std::vector<Record> records_;
  std::vector<std::vector<uint8_t>> raw_;
  std::vector<Parsed> parsed_;
};
DALI_REGISTER_OPERATOR(readers__TFRecord, TFRecordReaderOp, CPU);
DALI_REGISTER_OPERATOR(TFRecordReader, TFRecordReaderOp, CPU);

// =============================================================================================
// readers.mxnet (RecordIO)
// =============================================================================================
DALI_SCHEMA(readers__MXNet)
    .DocStr("Reads the data from an MXNet RecordIO: outputs (encoded image bytes, float label(s)).")
    .NumInput(0)
    .NumOutput(2)
    .AddArg("path", "List of paths to RecordIO files.", ArgType::STRING_VEC)
    .AddArg("index_path", "List (of length 1) with the path to the index file (lines `index offset`).", ArgType::STRING_VEC)
    .AddOptionalArg("use_o_direct", "Ignored.", ArgValue::Bool(false))
    .AddParent("LoaderBase");
DALI_SCHEMA(MXNetReader).DocStr("Legacy alias of readers.mxnet").NumInput(0).NumOutput(2).AddParent("readers__MXNet");

class MXNetReaderOp : public OperatorBase {
 public:
  explicit MXNetReaderOp(const OpSpec &spec) : OperatorBase(spec), loader_(spec), files_(StrVec(spec, "path")) {
    const auto index_paths = StrVec(spec, "index_path");
    DALI_ENFORCE(index_paths.size() == 1, "RecordIOReader supports only a single index file");
    std::vector<int64_t> file_offsets{0};
    for (int64_t s : files_.sizes) file_offsets.push_back(file_offsets.back() + s);
    std::ifstream idx(index_paths[0]);
    DALI_ENFORCE(idx.good(), "Could not open RecordIO index file. Provided path: \"", index_paths[0], "\"");
    std::vector<int64_t> offs;
    int64_t index, offset;
    while (idx >> index >> offset) offs.push_back(offset);
    DALI_ENFORCE(!offs.empty(), "RecordIO index file doesn't contain any indices. Provided path: \"", index_paths[0], "\"");
    std::sort(offs.begin(), offs.end());
    // the offsets run over the concatenation of the files; a record ends where the next one begins
    size_t fi = 0;
    for (size_t i = 0; i < offs.size(); i++) {
      while (fi + 1 < file_offsets.size() - 1 && offs[i] >= file_offsets[fi + 1]) fi++;
      const int64_t end = i + 1 < offs.size() ? std::min(offs[i + 1], file_offsets[fi + 1]) : file_offsets.back();
      const int64_t size = end - offs[i];
      if (size > 0) records_.push_back({offs[i] - file_offsets[fi], size, (int)fi});
    }
    DALI_ENFORCE(!records_.empty(), "RecordIO index describes no record");
    loader_.Init((int64_t)records_.size());
  }
  ReaderMeta GetReaderMeta() const override { return loader_.Meta(); }
  std::string SaveState() const override { return loader_.Save(); }
  void RestoreState(const std::string &s) override { loader_.Restore(s); }
  bool SetupImpl(std::vector<OutputDesc> &, const Workspace &) override { return false; }

  void RunImpl(Workspace &ws) override {
    const int n = max_batch_size_;
    std::vector<int64_t> picks(n);
    for (int i = 0; i < n; i++) picks[i] = loader_.NextIndex(i == 0);
    images_.resize(n);
    labels_.resize(n);
    for (int i = 0; i < n; i++) {
      ws.GetThreadPool().AddWork([this, &picks, i](int) {
        const Record &r = records_[picks[i]];
        std::vector<uint8_t> buf((size_t)r.size);
        files_.Read(r.file, r.offset, buf.data(), r.size);
        Parse(buf, make_string(files_.paths[r.file], " at index ", r.offset), &images_[i], &labels_[i]);
      }, records_[picks[i]].size);
    }
    ws.GetThreadPool().RunAll();
    std::vector<TensorShape> ishape(n), lshape(n);
    for (int i = 0; i < n; i++) { ishape[i] = {(int64_t)images_[i].size()}; lshape[i] = {(int64_t)labels_[i].size()}; }
    TensorList &img = ws.Output(0), &lab = ws.Output(1);
    img.Resize(ishape, DALI_UINT8);
    lab.Resize(lshape, DALI_FLOAT);
    img.source_info.resize(n);
    for (int i = 0; i < n; i++) {
      const Record &r = records_[picks[i]];
      img.source_info[i] = make_string(files_.paths[r.file], " at index ", r.offset);
      memcpy(img.raw(i), images_[i].data(), images_[i].size());
      memcpy(lab.raw(i), labels_[i].data(), labels_[i].size() * sizeof(float));
    }
  }

 private:
  // RecordIO record(s) -> image bytes + labels (recordio_parser.h:96-186): [magic][lrec = cflag << 29 | length]
  // [payload, padded to 4]; cflag 0 = whole record, 1 / 2 / 3 = first / middle / last segment of a record that
  // contained the magic number (which is re-inserted between segments).  Payload = ImageRecordIOHeader{flag, label,
  // id, id2} + `flag` extra float labels + the image.
  static void Parse(const std::vector<uint8_t> &buf, const std::string &where, std::vector<uint8_t> *image,
                    std::vector<float> *label) {
    constexpr uint32_t kMagic = 0xced7230a;
    const uint8_t *in = buf.data(), *end = buf.data() + buf.size();
    auto need = [&](size_t k, const char *what) {
      DALI_ENFORCE((size_t)(end - in) >= k, "Invalid RecordIO file: ", where, " (", what, " requires ", k, " bytes, but only ",
                   end - in, " bytes are available).");
    };
    auto u32 = [&](const char *what) { need(4, what); uint32_t v; memcpy(&v, in, 4); in += 4; return v; };
    DALI_ENFORCE(u32("magic number") == kMagic, "Invalid RecordIO: wrong magic number");
    uint32_t lrec = u32("length flag");
    uint32_t cflag = (lrec >> 29) & 7, clength = lrec & ((1u << 29) - 1);
    need(clength, "record payload");
    std::vector<uint8_t> data(in, in + clength);
    in += clength;
    while (cflag != 0 && cflag != 3) {
      const size_t pad = ((clength + 3) & ~3u) - clength;
      need(pad, "record padding");
      in += pad;
      const uint8_t m[4] = {0x0a, 0x23, 0xd7, 0xce};
      data.insert(data.end(), m, m + 4);
      DALI_ENFORCE(u32("segment magic number") == kMagic, "Invalid RecordIO file: ", where, " (wrong segment magic number).");
      lrec = u32("segment length flag");
      cflag = (lrec >> 29) & 7;
      clength = lrec & ((1u << 29) - 1);
      need(clength, "segment payload");
      data.insert(data.end(), in, in + clength);
      in += clength;
    }
    DALI_ENFORCE(data.size() >= 24, "Invalid RecordIO file: ", where, " (record payload length: ", data.size(),
                 " bytes, minimum is 24 bytes).");
    uint32_t flag;
    float first_label;
    memcpy(&flag, data.data(), 4);
    memcpy(&first_label, data.data() + 4, 4);
    const size_t label_bytes = (size_t)flag * sizeof(float);
    DALI_ENFORCE(label_bytes <= data.size() - 24, "Invalid RecordIO file: ", where, " (label size: ", label_bytes,
                 " bytes, available data: ", data.size() - 24, " bytes).");
    if (flag == 0) {
      label->assign(1, first_label);
    } else {
      label->resize(flag);
      memcpy(label->data(), data.data() + 24, label_bytes);
    }
    image->assign(data.begin() + 24 + label_bytes, data.end());
  }
  Loader loader_;
  OpenFiles files_;
  std::vector<Record> records_;
  std::vector<std::vector<uint8_t>> images_;
  std::vector<std::vector<float>> labels_;
};
DALI_REGISTER_OPERATOR(readers__MXNet, MXNetReaderOp, CPU);
DALI_REGISTER_OPERATOR(MXNetReader, MXNetReaderOp, CPU);

// =============================================================================================
// readers.webdataset
// =============================================================================================
DALI_SCHEMA(readers__Webdataset)
    .DocStr("A reader for the webdataset format: tar archives whose entries are grouped into samples by file name "
            "without extension; the extension sets in `ext` (alternatives separated by ';') select the components "
            "returned, one output each.  Index files (wds2idx, versions v1.1 / v1.2) are optional: without them the "
            "archives are scanned once at start-up.")
    .NumInput(0)
    .NumOutput(1)
    .AddArg("paths", "The list of (one or more) paths to the webdataset archives.", ArgType::STRING_VEC)
    .AddArg("ext", "The extension sets for each of the outputs produced (\"jpg;png\", \"cls\").", ArgType::STRING_VEC)
    .AddOptionalArg("case_sensitive_extensions", "Whether the extensions are matched case-sensitively.", ArgValue::Bool(true))
    .AddOptionalTypeArg("index_paths", "The index files of the archives (same length as `paths`).", ArgType::STRING_VEC)
    .AddOptionalArg("missing_component_behavior", "\"empty\" (default): an empty tensor; \"skip\": the sample is skipped; "
                    "\"error\": an exception.", ArgValue::Str(""))
    .AddOptionalTypeArg("dtypes", "Data types of the outputs (default: UINT8 for all).", ArgType::INT_VEC)
    .AddOptionalArg("shuffle_after_epoch", "Not supported (must be False).", ArgValue::Bool(false))
    .AddParent("LoaderBase");

class WebdatasetReaderOp : public OperatorBase {
 public:
  explicit WebdatasetReaderOp(const OpSpec &spec) : OperatorBase(spec), loader_(spec), files_(StrVec(spec, "paths")) {
    DALI_ENFORCE(!spec.GetBool("shuffle_after_epoch"), "readers.webdataset: shuffle_after_epoch is not supported");
    const bool cs = spec.GetBool("case_sensitive_extensions");
    auto lower = [&](std::string v) { if (!cs) for (auto &c : v) c = (char)tolower(c); return v; };
    for (auto &set : StrVec(spec, "ext")) {
      std::vector<std::string> alts;
      size_t pos = 0;
      while (pos <= set.size()) {
        size_t e = set.find(';', pos);
        if (e == std::string::npos) e = set.size();
        if (e > pos) alts.push_back(lower(set.substr(pos, e - pos)));
        pos = e + 1;
      }
      DALI_ENFORCE(!alts.empty(), "readers.webdataset: empty extension set");
      ext_.push_back(alts);
    }
    DALI_ENFORCE(!ext_.empty(), "readers.webdataset: `ext` must name at least one output");
    std::string mcb = spec.GetString("missing_component_behavior");
    if (mcb.empty()) mcb = "empty";
    for (auto &c : mcb) c = (char)tolower(c);
    DALI_ENFORCE(mcb == "empty" || mcb == "skip" || mcb == "error", "Invalid value for missing_component_behavior \"", mcb,
                 "\". Possible values are: empty, skip, error");
    dtypes_.assign(ext_.size(), DALI_UINT8);
    if (spec.TryArg("dtypes")) {
      auto dt = spec.GetIntVec("dtypes");
      DALI_ENFORCE(dt.size() == ext_.size(), "Number of extensions does not match the number of provided types");
      for (size_t i = 0; i < dt.size(); i++) dtypes_[i] = (DALIDataType)dt[i];
    }
    const auto index_paths = StrVec(spec, "index_paths");
    DALI_ENFORCE(index_paths.empty() || index_paths.size() == files_.paths.size(),
                 "The number of index files, if any, must match the number of archives in the dataset");
    for (size_t k = 0; k < files_.paths.size(); k++) {
      std::vector<Sample> found;
      if (index_paths.empty()) ScanTar((int)k, &found, lower); else ReadIndex((int)k, index_paths[k], &found, lower);
      for (auto &s : found) {
        bool complete = true;
        for (auto &c : s.comp) complete &= c.size >= 0;
        if (!complete && mcb == "skip") continue;
        DALI_ENFORCE(complete || mcb != "error", "Underful sample detected at ", files_.paths[k], ":", s.first_offset);
        bool any = false;
        for (auto &c : s.comp) any |= c.size >= 0;
        if (any) samples_.push_back(s);
      }
    }
    DALI_ENFORCE(!samples_.empty(), "No samples found in the webdataset archives for the requested extensions");
    loader_.Init((int64_t)samples_.size());
  }
  ReaderMeta GetReaderMeta() const override { return loader_.Meta(); }
  std::string SaveState() const override { return loader_.Save(); }
  void RestoreState(const std::string &s) override { loader_.Restore(s); }
  bool SetupImpl(std::vector<OutputDesc> &, const Workspace &) override { return false; }

  void RunImpl(Workspace &ws) override {
    const int n = max_batch_size_, no = (int)ext_.size();
    DALI_ENFORCE((int)ws.outputs.size() == no, "readers.webdataset: ", no, " extension sets but ", ws.outputs.size(), " outputs");
    std::vector<int64_t> picks(n);
    for (int i = 0; i < n; i++) picks[i] = loader_.NextIndex(i == 0);
    for (int o = 0; o < no; o++) {
      const int64_t esz = (int64_t)TypeSize(dtypes_[o]);
      std::vector<TensorShape> shapes(n);
      for (int i = 0; i < n; i++) {
        const Component &c = samples_[picks[i]].comp[o];
        const int64_t bytes = c.size < 0 ? 0 : c.size;
        DALI_ENFORCE(bytes % esz == 0, "Error in sample at ", files_.paths[samples_[picks[i]].file], ":", c.offset,
                     " - the size of a component (", bytes, " bytes) is not divisible by the size of the output type");
        shapes[i] = TensorShape{bytes / esz};
      }
      TensorList &out = ws.Output(o);
      out.Resize(shapes, dtypes_[o]);
      out.source_info.resize(n);
      for (int i = 0; i < n; i++) {
        const Sample &s = samples_[picks[i]];
        const Component &c = s.comp[o];
        out.source_info[i] = make_string("archive ", files_.paths[s.file], " component at ", c.offset);
        if (c.size > 0)
          ws.GetThreadPool().AddWork([this, &out, &s, &c, i](int) { files_.Read(s.file, c.offset, out.raw(i), c.size); }, c.size);
      }
    }
    ws.GetThreadPool().RunAll();
  }

 private:
  struct Component { int64_t offset = 0, size = -1; };  // size < 0: the sample has no such component
  struct Sample { int file = 0; int64_t first_offset = 0; std::vector<Component> comp; };

  // a component fills EVERY output whose extension set lists it (the ext -> outputs map of webdataset_loader.cc:413-460)
  void Assign(Sample *s, const std::string &ext, int64_t offset, int64_t size) const {
    for (size_t o = 0; o < ext_.size(); o++)
      for (auto &a : ext_[o])
        if (a == ext) {
          if (s->comp[o].size < 0) s->comp[o] = Component{offset, size};
          break;
        }
  }
  // base name / extension of a tar entry: the extension is the text behind the FIRST dot of the file name
  static bool SplitName(const std::string &path, std::string *base, std::string *ext) {
    const size_t slash = path.rfind('/');
    const size_t name = slash == std::string::npos ? 0 : slash + 1;
    if (name >= path.size() || path[name] == '.') return false;  // hidden files are not samples
    const size_t dot = path.find('.', name);
    if (dot == std::string::npos) return false;
    *base = path.substr(0, dot);
    *ext = path.substr(dot + 1);
    return true;
  }
  template <typename Lower>
  void AddEntry(std::vector<Sample> *out, std::string *cur_base, int file, const std::string &path, int64_t offset,
                int64_t size, Lower lower) const {
    std::string base, ext;
    if (!SplitName(path, &base, &ext)) return;
    if (out->empty() || base != *cur_base) {
      Sample s;
      s.file = file;
      s.first_offset = offset;
      s.comp.assign(ext_.size(), Component{});
      out->push_back(s);
      *cur_base = base;
    }
    Assign(&out->back(), lower(ext), offset, size);
  }
  // POSIX ustar / GNU tar: 512-byte headers, name [0,100) (+ prefix [345,500)), size octal [124,136), type flag [156];
  // GNU long names (type 'L') carry the name of the next entry; data is padded to 512 bytes; two zero blocks end it.
  template <typename Lower>
  void ScanTar(int file, std::vector<Sample> *out, Lower lower) const {
    int64_t pos = 0;
    std::string cur_base, long_name;
    uint8_t h[512];
    while (pos + 512 <= files_.sizes[file]) {
      files_.Read(file, pos, h, 512);
      bool zero = true;
      for (int i = 0; i < 512 && zero; i++) zero = h[i] == 0;
      if (zero) break;
      uint64_t usize = 0;
      bool size_ok = true;
      if (h[124] & 0x80) {  // base-256 (GNU): 11 more bytes, big endian; anything that does not fit 63 bits is malformed
        for (int i = 125; i < 136; i++) {
          size_ok = size_ok && (usize >> 55) == 0;
          usize = (usize << 8) | h[i];
        }
      } else {
        for (int i = 124; i < 136 && h[i] >= '0' && h[i] <= '7'; i++) usize = usize * 8 + (uint64_t)(h[i] - '0');  // <= 36 bits
      }
      const char type = (char)h[156];
      std::string name(reinterpret_cast<const char *>(h), strnlen(reinterpret_cast<const char *>(h), 100));
      if (!memcmp(h + 257, "ustar", 5) && h[345]) {
        std::string prefix(reinterpret_cast<const char *>(h + 345), strnlen(reinterpret_cast<const char *>(h + 345), 155));
        name = prefix + "/" + name;
      }
      const int64_t data = pos + 512;
      DALI_ENFORCE(size_ok && usize <= (uint64_t)(files_.sizes[file] - data), "Malformed tar archive ", files_.paths[file],
                   " (entry at ", pos, ")");
      const int64_t size = (int64_t)usize;
      if (type == 'L') {
        long_name.resize((size_t)size);
        files_.Read(file, data, &long_name[0], size);
        long_name.resize(strnlen(long_name.c_str(), long_name.size()));
      } else {
        if (!long_name.empty()) { name = long_name; long_name.clear(); }
        if (type == '0' || type == 0) AddEntry(out, &cur_base, file, name, data, size, lower);
      }
      pos = data + ((size + 511) & ~(int64_t)511);  // size >= 0: pos strictly increases
    }
  }
  // index file: "v1.2 <num_samples>" then one line per sample: "<ext> <data offset> <size> [<source name>] ..." (v1.2 adds
  // the file name to every component; v1.1 has triples only... both are told apart by the field count)
  template <typename Lower>
  void ReadIndex(int file, const std::string &path, std::vector<Sample> *out, Lower lower) const {
    std::ifstream idx(path);
    DALI_ENFORCE(idx.good(), "Could not open index file ", path);
    std::string version;
    int64_t count = 0;
    idx >> version >> count;
    DALI_ENFORCE(version == "v1.1" || version == "v1.2", "Unsupported version of the index file (", path, "): ", version);
    std::string line;
    std::getline(idx, line);
    for (int64_t k = 0; k < count; k++) {
      DALI_ENFORCE((bool)std::getline(idx, line), "Malformed index file at \"", path, "\" - expected ", count, " samples");
      std::istringstream ls(line);
      std::vector<std::string> tok;
      for (std::string t; ls >> t;) tok.push_back(t);
      const size_t per = version == "v1.2" ? 4 : 3;
      DALI_ENFORCE(!tok.empty() && tok.size() % per == 0, "Malformed index file at \"", path, "\" - sample line ", k);
      Sample s;
      s.file = file;
      s.comp.assign(ext_.size(), Component{});
      for (size_t c = 0; c < tok.size(); c += per) {
        const int64_t off = std::stoll(tok[c + 1]), size = std::stoll(tok[c + 2]);
        DALI_ENFORCE(off >= 0 && size >= 0 && off <= files_.sizes[file] && size <= files_.sizes[file] - off, "Index file ", path, " does not describe ",
                     files_.paths[file]);
        if (c == 0) s.first_offset = off;
        Assign(&s, lower(tok[c]), off, size);
      }
      out->push_back(s);
    }
  }

  Loader loader_;
  OpenFiles files_;
  std::vector<std::vector<std::string>> ext_;
  std::vector<DALIDataType> dtypes_;
  std::vector<Sample> samples_;
};
DALI_REGISTER_OPERATOR(readers__Webdataset, WebdatasetReaderOp, CPU);

}  // namespace daliamd_host
