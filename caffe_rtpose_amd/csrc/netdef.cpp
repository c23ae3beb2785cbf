// netdef.cpp — graph builder, prototxt text parser/emitter, .caffemodel reader/writer, synthetic weights.
#include "netdef.h"

#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <sstream>

namespace rtp {

// ---------------------------------------------------------------------------------------
// Built-in linevec graphs
// ---------------------------------------------------------------------------------------
static void add_conv(NetDef& n, const std::string& name, const std::string& bottom, int cout, int k, bool relu) {
  LayerDef c;
  c.name = name; c.type = "Convolution"; c.bottoms = {bottom}; c.tops = {name};
  c.num_output = cout; c.kernel = k; c.pad = (k - 1) / 2; c.stride = 1;
  n.layers.push_back(c);
  if (relu) {
    LayerDef r;
    // the reference names them relu1_1, relu4_3_CPM, Mrelu1_stage2_L1, ... ; names carry no weights
    std::string rn = name;
    if (rn.rfind("Mconv", 0) == 0) rn = "Mrelu" + rn.substr(5);
    else if (rn.rfind("conv", 0) == 0) rn = "relu" + rn.substr(4);
    r.name = rn; r.type = "ReLU"; r.bottoms = {name}; r.tops = {name};
    n.layers.push_back(r);
  }
}
static void add_pool(NetDef& n, const std::string& name, const std::string& bottom) {
  LayerDef p;
  p.name = name; p.type = "Pooling"; p.bottoms = {bottom}; p.tops = {name};
  p.pool_method = "MAX"; p.pool_kernel = 2; p.pool_stride = 2;
  n.layers.push_back(p);
}
static void add_concat(NetDef& n, const std::string& name, std::vector<std::string> bottoms) {
  LayerDef c;
  c.name = name; c.type = "Concat"; c.bottoms = bottoms; c.tops = {name}; c.axis = 1;
  n.layers.push_back(c);
}

NetDef build_linevec(int model) {
  NetDef n;
  const bool coco = model == 0;
  const int npaf = coco ? 38 : 28, nheat = coco ? 19 : 16;
  n.inputs = {"image"};
  n.input_dim = {1, 3, 540, 960};  // placeholder, reshaped at run time (rtpose.cpp:188-191)
  add_conv(n, "conv1_1", "image", 64, 3, true);
  add_conv(n, "conv1_2", "conv1_1", 64, 3, true);
  add_pool(n, "pool1_stage1", "conv1_2");
  add_conv(n, "conv2_1", "pool1_stage1", 128, 3, true);
  add_conv(n, "conv2_2", "conv2_1", 128, 3, true);
  add_pool(n, "pool2_stage1", "conv2_2");
  add_conv(n, "conv3_1", "pool2_stage1", 256, 3, true);
  add_conv(n, "conv3_2", "conv3_1", 256, 3, true);
  add_conv(n, "conv3_3", "conv3_2", 256, 3, true);
  add_conv(n, "conv3_4", "conv3_3", 256, 3, true);
  add_pool(n, "pool3_stage1", "conv3_4");
  add_conv(n, "conv4_1", "pool3_stage1", 512, 3, true);
  add_conv(n, "conv4_2", "conv4_1", 512, 3, true);
  add_conv(n, "conv4_3_CPM", "conv4_2", 256, 3, true);
  add_conv(n, "conv4_4_CPM", "conv4_3_CPM", 128, 3, true);
  const char* L[2] = {"L1", "L2"};
  const int lout[2] = {npaf, nheat};
  char nm[64], bt[64];
  for (int i = 1; i <= 5; ++i)
    for (int l = 0; l < 2; ++l) {
      snprintf(nm, sizeof nm, "conv5_%d_CPM_%s", i, L[l]);
      if (i == 1) snprintf(bt, sizeof bt, "conv4_4_CPM");
      else snprintf(bt, sizeof bt, "conv5_%d_CPM_%s", i - 1, L[l]);
      if (i <= 3) add_conv(n, nm, bt, 128, 3, true);
      else if (i == 4) add_conv(n, nm, bt, 512, 1, true);
      else add_conv(n, nm, bt, lout[l], 1, false);
    }
  std::string p1 = "conv5_5_CPM_L1", p2 = "conv5_5_CPM_L2";
  for (int s = 2; s <= 6; ++s) {
    snprintf(nm, sizeof nm, "concat_stage%d", s);
    const std::string cc = nm;
    add_concat(n, cc, {p1, p2, "conv4_4_CPM"});
    for (int i = 1; i <= 7; ++i)
      for (int l = 0; l < 2; ++l) {
        snprintf(nm, sizeof nm, "Mconv%d_stage%d_%s", i, s, L[l]);
        if (i == 1) snprintf(bt, sizeof bt, "%s", cc.c_str());
        else snprintf(bt, sizeof bt, "Mconv%d_stage%d_%s", i - 1, s, L[l]);
        if (i <= 5) add_conv(n, nm, bt, 128, 7, true);
        else if (i == 6) add_conv(n, nm, bt, 128, 1, true);
        else add_conv(n, nm, bt, lout[l], 1, false);
      }
    snprintf(nm, sizeof nm, "Mconv7_stage%d_L1", s); p1 = nm;
    snprintf(nm, sizeof nm, "Mconv7_stage%d_L2", s); p2 = nm;
  }
  add_concat(n, "concat_stage7", {p2, p1});
  LayerDef rz;
  rz.name = "resize"; rz.type = "ImResize"; rz.bottoms = {"concat_stage7"}; rz.tops = {"resized_map"};
  rz.factor = 8; rz.scale_gap = 0.3f; rz.start_scale = 1.0f;
  n.layers.push_back(rz);
  LayerDef nms;
  nms.name = "nms"; nms.type = "Nms"; nms.bottoms = {"resized_map"}; nms.tops = {"joints"};
  if (coco) { nms.nms_threshold = 0.05f; nms.max_peaks = 64; nms.num_parts = 18; }
  else { nms.nms_threshold = 0.6f; nms.max_peaks = 20; nms.num_parts = 15; }
  n.layers.push_back(nms);
  n.name = coco ? "linevec_coco18" : "linevec_mpi15";
  return n;
}

// ---------------------------------------------------------------------------------------
// protobuf TEXT format (just enough for deploy prototxts)
// ---------------------------------------------------------------------------------------
struct PMsg;
struct PField {
  std::string key;
  std::string scalar;  // if !msg
  PMsg* msg = nullptr;
};
struct PMsg {
  std::vector<PField> f;
  ~PMsg() { for (auto& x : f) delete x.msg; }
  const PMsg* sub(const char* k) const { for (auto& x : f) if (x.key == k && x.msg) return x.msg; return nullptr; }
  bool get(const char* k, std::string* v) const { for (auto& x : f) if (x.key == k && !x.msg) { *v = x.scalar; return true; } return false; }
  std::vector<std::string> all(const char* k) const { std::vector<std::string> r; for (auto& x : f) if (x.key == k && !x.msg) r.push_back(x.scalar); return r; }
};

struct Lexer {
  const std::string& s;
  size_t p = 0;
  explicit Lexer(const std::string& t) : s(t) {}
  void skip() {
    while (p < s.size()) {
      if (s[p] == '#') { while (p < s.size() && s[p] != '\n') ++p; }
      else if (isspace((unsigned char)s[p]) || s[p] == ',' || s[p] == ';') ++p;
      else break;
    }
  }
  // returns token type: 0 eof, 1 ident/number, 2 string, '{','}',':','<','>'
  int next(std::string* tok) {
    skip();
    if (p >= s.size()) return 0;
    const char c = s[p];
    if (c == '{' || c == '}' || c == ':' || c == '<' || c == '>') { ++p; return c; }
    if (c == '"' || c == '\'') {
      const char q = c;
      ++p;
      tok->clear();
      while (p < s.size() && s[p] != q) {
        if (s[p] == '\\' && p + 1 < s.size()) { ++p; }
        tok->push_back(s[p++]);
      }
      ++p;
      return 2;
    }
    tok->clear();
    while (p < s.size() && !isspace((unsigned char)s[p]) && !strchr("{}:<>#\"',;", s[p])) tok->push_back(s[p++]);
    return 1;
  }
};

static bool parse_msg(Lexer& lx, PMsg* m, bool top, std::string* err) {
  std::string tok;
  for (;;) {
    const int t = lx.next(&tok);
    if (t == 0) { if (top) return true; *err = "unexpected end of file inside a message"; return false; }
    if (t == '}' || t == '>') { if (top) { *err = "unbalanced '}'"; return false; } return true; }
    if (t != 1) { *err = "expected a field name near offset " + std::to_string(lx.p); return false; }
    PField f;
    f.key = tok;
    int t2 = lx.next(&tok);
    if (t2 == ':') {
      t2 = lx.next(&tok);
      if (t2 == '{' || t2 == '<') {
        f.msg = new PMsg();
        if (!parse_msg(lx, f.msg, false, err)) { delete f.msg; return false; }
      } else if (t2 == 1 || t2 == 2) {
        f.scalar = tok;
      } else { *err = "expected a value for '" + f.key + "'"; return false; }
    } else if (t2 == '{' || t2 == '<') {
      f.msg = new PMsg();
      if (!parse_msg(lx, f.msg, false, err)) { delete f.msg; return false; }
    } else { *err = "expected ':' or '{' after '" + f.key + "'"; return false; }
    m->f.push_back(f);
    f.msg = nullptr;
  }
}

static int to_i(const std::string& s) { return (int)strtol(s.c_str(), nullptr, 10); }
static float to_f(const std::string& s) { return strtof(s.c_str(), nullptr); }

static std::string v1_type(const std::string& t) {
  // V1LayerParameter.LayerType enum names (caffe.proto V1LayerParameter) -> new-style type strings
  if (t == "CONVOLUTION") return "Convolution";
  if (t == "RELU") return "ReLU";
  if (t == "POOLING") return "Pooling";
  if (t == "CONCAT") return "Concat";
  if (t == "SPLIT") return "Split";
  return t;
}

bool parse_prototxt(const std::string& text, NetDef* out, std::string* err) {
  Lexer lx(text);
  PMsg root;
  if (!parse_msg(lx, &root, true, err)) return false;
  NetDef n;
  root.get("name", &n.name);
  n.inputs = root.all("input");
  for (auto& d : root.all("input_dim")) n.input_dim.push_back(to_i(d));
  for (auto& fld : root.f) {
    if (!fld.msg) continue;
    if (fld.key == "input_shape") { for (auto& d : fld.msg->all("dim")) n.input_dim.push_back(to_i(d)); continue; }
    if (fld.key != "layer" && fld.key != "layers") continue;
    const PMsg& m = *fld.msg;
    LayerDef L;
    std::string v;
    m.get("name", &L.name);
    if (m.get("type", &v)) L.type = v1_type(v);
    L.bottoms = m.all("bottom");
    L.tops = m.all("top");
    if (const PMsg* c = m.sub("convolution_param")) {
      if (c->get("num_output", &v)) L.num_output = to_i(v);
      if (c->get("kernel_size", &v)) L.kernel = to_i(v);
      if (c->get("pad", &v)) L.pad = to_i(v);
      if (c->get("stride", &v)) L.stride = to_i(v);
      if (c->get("bias_term", &v)) L.bias_term = (v == "true" || v == "1");
      std::string kh, kw, g, dl;
      const bool has_kh = c->get("kernel_h", &kh), has_kw = c->get("kernel_w", &kw);  // Caffe accepts kernel_size or the h/w pair
      if (has_kh || has_kw) {
        if (!(has_kh && has_kw) || to_i(kh) != to_i(kw)) { *err = "layer " + L.name + ": non-square kernels are outside the linevec path"; return false; }
        L.kernel = to_i(kh);
      }
      if (c->get("group", &g) && to_i(g) != 1) { *err = "layer " + L.name + ": group != 1 is outside the linevec path"; return false; }
      if (c->get("dilation", &dl) && to_i(dl) != 1) { *err = "layer " + L.name + ": dilation != 1 is outside the linevec path"; return false; }
    }
    if (const PMsg* c = m.sub("pooling_param")) {
      if (c->get("pool", &v)) L.pool_method = v;
      if (c->get("kernel_size", &v)) L.pool_kernel = to_i(v);
      if (c->get("stride", &v)) L.pool_stride = to_i(v);
      if (c->get("pad", &v)) L.pool_pad = to_i(v);
    }
    if (const PMsg* c = m.sub("relu_param")) { if (c->get("negative_slope", &v)) L.negative_slope = to_f(v); }
    if (const PMsg* c = m.sub("concat_param")) {
      if (c->get("axis", &v)) L.axis = to_i(v);
      if (c->get("concat_dim", &v)) L.axis = to_i(v);
    }
    if (const PMsg* c = m.sub("imresize_param")) {
      if (c->get("target_spatial_width", &v)) L.target_w = to_i(v);
      if (c->get("target_spatial_height", &v)) L.target_h = to_i(v);
      if (c->get("factor", &v)) L.factor = to_f(v);
      if (c->get("start_scale", &v)) L.start_scale = to_f(v);
      if (c->get("scale_gap", &v)) L.scale_gap = to_f(v);
    }
    if (const PMsg* c = m.sub("nms_param")) {
      if (c->get("threshold", &v)) L.nms_threshold = to_f(v);
      if (c->get("max_peaks", &v)) L.max_peaks = to_i(v);
      if (c->get("num_parts", &v)) L.num_parts = to_i(v);
    }
    if (L.type == "Input") {
      if (const PMsg* ip = m.sub("input_param"))
        if (const PMsg* sh = ip->sub("shape")) for (auto& d : sh->all("dim")) n.input_dim.push_back(to_i(d));
      for (auto& t : L.tops) n.inputs.push_back(t);
      continue;
    }
    n.layers.push_back(L);
  }
  if (n.layers.empty()) { *err = "no layers found"; return false; }
  *out = n;
  return true;
}

std::string emit_prototxt(const NetDef& net) {
  std::ostringstream o;
  if (!net.name.empty()) o << "name: \"" << net.name << "\"\n";
  for (auto& i : net.inputs) o << "input: \"" << i << "\"\n";
  for (int d : net.input_dim) o << "input_dim: " << d << "\n";
  for (auto& L : net.layers) {
    o << "layer {\n  name: \"" << L.name << "\"\n  type: \"" << L.type << "\"\n";
    for (auto& b : L.bottoms) o << "  bottom: \"" << b << "\"\n";
    for (auto& t : L.tops) o << "  top: \"" << t << "\"\n";
    if (L.type == "Convolution") {
      o << "  convolution_param {\n    num_output: " << L.num_output << "\n    pad: " << L.pad << "\n    kernel_size: " << L.kernel << "\n";
      if (L.stride != 1) o << "    stride: " << L.stride << "\n";
      o << "  }\n";
    } else if (L.type == "Pooling") {
      o << "  pooling_param {\n    pool: " << L.pool_method << "\n    kernel_size: " << L.pool_kernel << "\n    stride: " << L.pool_stride << "\n  }\n";
    } else if (L.type == "Concat") {
      o << "  concat_param {\n    axis: " << L.axis << "\n  }\n";
    } else if (L.type == "ImResize") {
      o << "  imresize_param {\n    factor: " << L.factor << "\n    scale_gap: " << L.scale_gap << "\n    start_scale: " << L.start_scale << "\n  }\n";
    } else if (L.type == "Nms") {
      o << "  nms_param {\n    threshold: " << L.nms_threshold << "\n    max_peaks: " << L.max_peaks << "\n    num_parts: " << L.num_parts << "\n  }\n";
    }
    o << "}\n";
  }
  return o.str();
}

// ---------------------------------------------------------------------------------------
// protobuf WIRE format
// ---------------------------------------------------------------------------------------
struct Rd {
  const uint8_t* p;
  const uint8_t* e;
  bool ok = true;
  uint64_t varint() {
    uint64_t v = 0;
    int sh = 0;
    while (p < e) {
      const uint8_t b = *p++;
      v |= (uint64_t)(b & 0x7f) << sh;
      if (!(b & 0x80)) return v;
      sh += 7;
      if (sh > 63) break;
    }
    ok = false;
    return 0;
  }
  bool skip(int wt) {
    if (wt == 0) { varint(); return ok; }
    if (wt == 1) { if (e - p < 8) return ok = false; p += 8; return true; }
    if (wt == 5) { if (e - p < 4) return ok = false; p += 4; return true; }
    if (wt == 2) { const uint64_t n = varint(); if (!ok || (uint64_t)(e - p) < n) return ok = false; p += n; return true; }
    return ok = false;
  }
};

static bool parse_blob(Rd r, BlobData* b) {
  int64_t legacy[4] = {0, 0, 0, 0};
  bool has_legacy = false;
  while (r.p < r.e && r.ok) {
    const uint64_t key = r.varint();
    const int fn = (int)(key >> 3), wt = (int)(key & 7);
    if (fn == 7 && wt == 2) {  // BlobShape
      const uint64_t n = r.varint();
      if (!r.ok || (uint64_t)(r.e - r.p) < n) return false;
      Rd s{r.p, r.p + n};
      r.p += n;
      while (s.p < s.e && s.ok) {
        const uint64_t k2 = s.varint();
        if ((k2 >> 3) == 1 && (k2 & 7) == 2) {
          const uint64_t m = s.varint();
          if (!s.ok || (uint64_t)(s.e - s.p) < m || m > 64) return false;  // packed dims stay inside the BlobShape; a blob has a handful of axes
          Rd d{s.p, s.p + m};
          s.p += m;
          while (d.p < d.e && d.ok) b->shape.push_back((int64_t)d.varint());
          if (!d.ok) return false;
        } else if ((k2 >> 3) == 1 && (k2 & 7) == 0) b->shape.push_back((int64_t)s.varint());
        else if (!s.skip((int)(k2 & 7))) return false;
      }
    } else if (fn == 5 && wt == 2) {  // packed floats
      const uint64_t n = r.varint();
      if (!r.ok || (uint64_t)(r.e - r.p) < n || (n & 3)) return false;
      const size_t old = b->data.size();
      b->data.resize(old + n / 4);
      if (n) memcpy(b->data.data() + old, r.p, n);  // an empty packed field is legal; memcpy(NULL, .., 0) is not
      r.p += n;
    } else if (fn == 5 && wt == 5) {  // unpacked float
      if (r.e - r.p < 4) return false;
      float v;
      memcpy(&v, r.p, 4);
      r.p += 4;
      b->data.push_back(v);
    } else if (fn >= 1 && fn <= 4 && wt == 0) {
      legacy[fn - 1] = (int64_t)r.varint();
      has_legacy = true;
    } else if (!r.skip(wt)) return false;
  }
  if (b->shape.empty() && has_legacy) b->shape.assign(legacy, legacy + 4);
  return r.ok;
}

static bool parse_layer(Rd r, bool v1, LayerWeights* L) {
  const int f_name = v1 ? 4 : 1, f_blobs = v1 ? 6 : 7;
  while (r.p < r.e && r.ok) {
    const uint64_t key = r.varint();
    const int fn = (int)(key >> 3), wt = (int)(key & 7);
    if (wt == 2) {
      const uint64_t n = r.varint();
      if (!r.ok || (uint64_t)(r.e - r.p) < n) return false;
      if (fn == f_name) L->name.assign((const char*)r.p, n);
      else if (!v1 && fn == 2) L->type.assign((const char*)r.p, n);
      else if (fn == f_blobs) {
        BlobData b;
        if (!parse_blob(Rd{r.p, r.p + n}, &b)) return false;
        L->blobs.push_back(std::move(b));
      }
      r.p += n;
    } else if (!r.skip(wt)) return false;
  }
  return r.ok;
}

bool read_caffemodel(const std::string& path, std::vector<LayerWeights>* out, std::string* err) {
  std::ifstream f(path, std::ios::binary);
  if (!f) { *err = "cannot open " + path; return false; }
  std::vector<uint8_t> buf((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
  Rd r{buf.data(), buf.data() + buf.size()};
  while (r.p < r.e && r.ok) {
    const uint64_t key = r.varint();
    const int fn = (int)(key >> 3), wt = (int)(key & 7);
    if ((fn == 100 || fn == 2) && wt == 2) {
      const uint64_t n = r.varint();
      if (!r.ok || (uint64_t)(r.e - r.p) < n) { r.ok = false; break; }
      LayerWeights L;
      if (!parse_layer(Rd{r.p, r.p + n}, fn == 2, &L)) { r.ok = false; break; }
      out->push_back(std::move(L));
      r.p += n;
    } else if (!r.skip(wt)) break;
  }
  if (!r.ok) { *err = "malformed NetParameter in " + path; return false; }
  return true;
}

static void put_varint(std::string& s, uint64_t v) {
  while (v >= 0x80) { s.push_back((char)((v & 0x7f) | 0x80)); v >>= 7; }
  s.push_back((char)v);
}
static void put_key(std::string& s, int fn, int wt) { put_varint(s, ((uint64_t)fn << 3) | wt); }
static void put_bytes(std::string& s, int fn, const std::string& b) { put_key(s, fn, 2); put_varint(s, b.size()); s += b; }

bool write_caffemodel(const std::string& path, const std::string& net_name, const std::vector<LayerWeights>& layers,
                      std::string* err) {
  std::ofstream f(path, std::ios::binary);
  if (!f) { *err = "cannot create " + path; return false; }
  std::string top;
  put_bytes(top, 1, net_name);
  f.write(top.data(), top.size());
  for (auto& L : layers) {
    std::string lm;
    put_bytes(lm, 1, L.name);
    put_bytes(lm, 2, L.type);
    for (auto& b : L.blobs) {
      std::string bm, sh, dims;
      for (int64_t d : b.shape) put_varint(dims, (uint64_t)d);
      put_bytes(sh, 1, dims);
      put_bytes(bm, 7, sh);
      put_key(bm, 5, 2);
      put_varint(bm, b.data.size() * 4);
      bm.append((const char*)b.data.data(), b.data.size() * 4);
      put_bytes(lm, 7, bm);
    }
    std::string hdr;
    put_key(hdr, 100, 2);
    put_varint(hdr, lm.size());
    f.write(hdr.data(), hdr.size());
    f.write(lm.data(), lm.size());
  }
  return (bool)f;
}

// ---------------------------------------------------------------------------------------
// synthetic weights
// ---------------------------------------------------------------------------------------
static inline uint64_t splitmix64(uint64_t& s) {
  s += 0x9E3779B97F4A7C15ull;
  uint64_t z = s;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
static inline uint64_t fnv1a(const std::string& s) {
  uint64_t h = 0xcbf29ce484222325ull;
  for (unsigned char c : s) { h ^= c; h *= 0x100000001b3ull; }
  return h;
}
void synth_conv_weights(uint64_t seed, const std::string& layer_name, int cout, int cin, int k, std::vector<float>* w,
                        std::vector<float>* b) {
  uint64_t s = seed ^ fnv1a(layer_name);
  const long nw = (long)cout * cin * k * k;
  w->resize(nw);
  b->resize(cout);
  const double stdv = std::sqrt(2.0 / ((double)cin * k * k));
  for (long i = 0; i < nw; ++i) {
    // Irwin-Hall: sum of 12 U(0,1) - 6 ~ N(0,1); each uniform is a 24-bit integer / 2^24 (exact)
    uint64_t acc = 0;
    for (int j = 0; j < 6; ++j) {
      const uint64_t r = splitmix64(s);
      acc += (r >> 40) + ((r >> 8) & 0xFFFFFFull);
    }
    const double z = (double)acc / 16777216.0 - 6.0;
    (*w)[i] = (float)(z * stdv);
  }
  for (int i = 0; i < cout; ++i) {
    const uint64_t r = splitmix64(s);
    (*b)[i] = (float)(((double)(r >> 40) / 16777216.0) * 0.2 - 0.1);
  }
}

}  // namespace rtp
