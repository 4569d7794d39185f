// dbhip_host.hpp — C++ host mirror of the reference's operator surface on top of the C-ABI.
//
// The reference's host side is Rust (not available in this image), so the layer that a Databend
// maintainer would write inside `ScalarFunction::eval` / `Transform` / `AccumulatingTransform` /
// `Join` impls is restated here in C++17 with the SAME vocabulary, argument
// meaning and error behaviour, so that tests read like the reference's own:
//   DataType / Scalar / Column / Value / DataBlock   src/query/expression/src/{types.rs,values.rs,block.rs:49-59}
//   FunctionSignature / Function / FunctionRegistry  src/query/expression/src/function.rs:87-134,275-379
//   EvalContext::set_error / render_error            src/query/expression/src/function.rs:534-620
//   Expr / Evaluator::run / partial_run              src/query/expression/src/{expression.rs,evaluator.rs:229-464}
//   FilterExecutor                                   src/query/expression/src/filter/filter_executor.rs:81-118
//   AggregateHashTable                               src/query/expression/src/aggregate/aggregate_hashtable.rs:42-519
//   Transform / AccumulatingTransform                src/query/pipeline/transforms/src/processors/transforms/{transform.rs:30-48,transform_accumulating.rs:30-38}
//   TransformPartialAggregate / TransformFinalAggregate / PartialSingleStateAggregator
//                                                    src/query/service/src/pipelines/processors/transforms/aggregator/*.rs
//   Join / JoinStream / InnerHashJoin                src/query/service/src/pipelines/processors/transforms/new_hash_join/{join.rs:22-53,memory/inner_join.rs}
//   DataBlock::sort                                  src/query/expression/src/kernels/sort.rs:91-113
// Every compute step goes through libdbhip.so; this file contains no arithmetic on column data.
#pragma once
#include <cstdint>
#include <cstring>
#include <functional>
#include <map>
#include <memory>
#include <optional>
#include <set>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/dbhip.h"

namespace dbhip_host {

// ---- errors (ErrorCode, src/common/exception) ------------------------------------------------
struct ErrorCode : std::runtime_error {
  std::string kind;  // "BadArguments", "Overflow", "Unimplemented", "Internal"
  ErrorCode(std::string k, const std::string& m) : std::runtime_error(m), kind(std::move(k)) {}
  static ErrorCode BadArguments(const std::string& m) { return ErrorCode("BadArguments", m); }
  static ErrorCode Overflow(const std::string& m) { return ErrorCode("Overflow", m); }
  static ErrorCode Unimplemented(const std::string& m) { return ErrorCode("Unimplemented", m); }
  static ErrorCode Internal(const std::string& m) { return ErrorCode("Internal", m); }
};

inline void check(int32_t rc) {
  if (rc == DBHIP_OK) return;
  std::string m = dbhip_last_error();
  if (rc == DBHIP_ERR_OVERFLOW) throw ErrorCode::Overflow(m);
  if (rc == DBHIP_ERR_UNSUPPORTED) throw ErrorCode::Unimplemented(m);
  if (rc == DBHIP_ERR_INVALID) throw ErrorCode::BadArguments(m);
  throw ErrorCode::Internal(m);
}

inline void init(int device = 0) { check(dbhip_init(device)); }

// ---- device memory ---------------------------------------------------------------------------
class DeviceBuffer {
 public:
  explicit DeviceBuffer(size_t bytes) : bytes_(bytes) { check(dbhip_alloc(bytes < 16 ? 16 : bytes + 16, &p_)); }
  ~DeviceBuffer() { if (p_) dbhip_free(p_); }
  DeviceBuffer(const DeviceBuffer&) = delete;
  DeviceBuffer& operator=(const DeviceBuffer&) = delete;
  void* ptr() const { return p_; }
  size_t bytes() const { return bytes_; }
  void upload(const void* src, size_t n) { if (n) check(dbhip_memcpy_h2d(p_, src, n, nullptr)); }
  void download(void* dst, size_t n) const { if (n) check(dbhip_memcpy_d2h(dst, p_, n, nullptr)); }
  void fill(int byte) { check(dbhip_memset(p_, byte, bytes_ < 16 ? 16 : bytes_, nullptr)); }
 private:
  void* p_ = nullptr;
  size_t bytes_;
};
using Buf = std::shared_ptr<DeviceBuffer>;
inline Buf make_buf(size_t bytes) { return std::make_shared<DeviceBuffer>(bytes); }

// ---- types -----------------------------------------------------------------------------------
struct DataType {
  int32_t id = DBHIP_T_I64;  // dbhip_type
  bool nullable = false;
  uint8_t precision = 0, scale = 0;  // decimals
  int32_t dim = 0;                   // Vector(dim) when id == DBHIP_T_F32 && dim > 0

  static DataType of(int32_t id, bool nullable = false) { DataType t; t.id = id; t.nullable = nullable; return t; }
  static DataType Decimal(uint8_t p, uint8_t s, bool nullable = false) {
    DataType t; t.id = p <= 18 ? DBHIP_T_DEC64 : DBHIP_T_DEC128; t.precision = p; t.scale = s; t.nullable = nullable; return t;
  }
  static DataType Vector(int32_t dim) { DataType t; t.id = DBHIP_T_F32; t.dim = dim; return t; }
  DataType wrap_nullable() const { DataType t = *this; t.nullable = true; return t; }
  DataType remove_nullable() const { DataType t = *this; t.nullable = false; return t; }
  bool is_decimal() const { return id == DBHIP_T_DEC64 || id == DBHIP_T_DEC128; }
  bool is_numeric() const { return id >= DBHIP_T_I8 && id <= DBHIP_T_F64; }
  bool same_physical(const DataType& o) const { return id == o.id && precision == o.precision && scale == o.scale && dim == o.dim; }
  size_t elem_size() const {
    switch (id) {
      case DBHIP_T_I8: case DBHIP_T_U8: return 1;
      case DBHIP_T_I16: case DBHIP_T_U16: return 2;
      case DBHIP_T_I32: case DBHIP_T_U32: case DBHIP_T_F32: case DBHIP_T_DATE: return dim > 0 ? 4 * (size_t)dim : 4;
      case DBHIP_T_DEC128: case DBHIP_T_STRING: return 16;
      case DBHIP_T_BOOL: return 0;
      default: return 8;
    }
  }
  std::string name() const {
    static const char* n[] = {"?", "Boolean", "Int8", "Int16", "Int32", "Int64", "UInt8", "UInt16", "UInt32", "UInt64", "Float32",
                              "Float64", "Date", "Timestamp", "Decimal", "Decimal", "String"};
    std::string s = dim > 0 ? "Vector(" + std::to_string(dim) + ")" : n[id];
    if (is_decimal()) s += "(" + std::to_string(precision) + ", " + std::to_string(scale) + ")";
    return nullable ? s + " NULL" : s;
  }
};

// Scalar (values.rs:122 Value::Scalar): numbers, decimals, booleans, dates; NULL.
struct Scalar {
  DataType type;
  bool is_null = false;
  int64_t i = 0;
  uint64_t u = 0;
  double f = 0;
  __int128 dec = 0;
  static Scalar Int(int32_t id, int64_t v) { Scalar s; s.type = DataType::of(id); s.i = v; s.u = (uint64_t)v; s.f = (double)v; s.dec = v; return s; }
  static Scalar Float(int32_t id, double v) { Scalar s; s.type = DataType::of(id); s.f = v; return s; }
  static Scalar Dec(uint8_t p, uint8_t sc, __int128 v) { Scalar s; s.type = DataType::Decimal(p, sc); s.dec = v; s.i = (int64_t)v; return s; }
  static Scalar Null(DataType t) { Scalar s; s.type = t.wrap_nullable(); s.is_null = true; return s; }
  // raw little-endian image in the column's physical type
  void store(void* out) const {
    switch (type.id) {
      case DBHIP_T_F32: { float x = (float)f; memcpy(out, &x, 4); } break;
      case DBHIP_T_F64: memcpy(out, &f, 8); break;
      case DBHIP_T_DEC128: memcpy(out, &dec, 16); break;
      case DBHIP_T_DEC64: { int64_t x = (int64_t)dec; memcpy(out, &x, 8); } break;
      case DBHIP_T_BOOL: { uint8_t x = i != 0; memcpy(out, &x, 1); } break;
      default: memcpy(out, &i, type.elem_size()); break;
    }
  }
  std::string to_string() const {
    if (is_null) return "NULL";
    if (type.id == DBHIP_T_F32 || type.id == DBHIP_T_F64) { std::ostringstream o; o << f; return o.str(); }
    if (type.id == DBHIP_T_U64) return std::to_string(u);
    return std::to_string(type.is_decimal() ? (long long)dec : (long long)i);
  }
};

// Column (values.rs:192): values + optional validity Bitmap, resident in HBM.
struct Column {
  DataType type;
  int64_t len = 0;
  Buf data;       // numbers: T[len]; Boolean: LSB-first bits; String: 16-byte views; Vector: f32[len][dim]
  Buf validity;   // LSB-first bits, NULL when the column is not nullable / has no NULLs
  Buf str_data;   // String: data buffer 0 (long strings)
  Buf str_ptrs;   // String: device array of device pointers to the data buffers
  bool is_const = false;  // BlockEntry::Const (block.rs): ONE stored value that stands for every row (e.g. a query vector)

  template <typename T>
  static Column from_vector(DataType t, const std::vector<T>& v, const std::vector<bool>* valid = nullptr) {
    Column c; c.type = t; c.len = (int64_t)(t.dim > 0 ? v.size() / t.dim : v.size());
    c.data = make_buf(v.size() * sizeof(T));
    c.data->upload(v.data(), v.size() * sizeof(T));
    if (valid) { c.type.nullable = true; c.validity = pack_bits(*valid); }
    return c;
  }
  // a one-row column holding a scalar (the operand form of a constant: dbhip_col.is_scalar)
  static Column from_scalar(const Scalar& sc) {
    Column c; c.type = sc.type; c.len = 1;
    if (sc.type.id == DBHIP_T_BOOL) { c.data = pack_bits(std::vector<bool>{sc.i != 0}); return c; }
    uint8_t raw[32] = {0};
    sc.store(raw);
    c.data = make_buf(32);
    c.data->upload(raw, 32);
    return c;
  }
  static Column from_bools(const std::vector<bool>& v, const std::vector<bool>* valid = nullptr) {
    Column c; c.type = DataType::of(DBHIP_T_BOOL); c.len = (int64_t)v.size(); c.data = pack_bits(v);
    if (valid) { c.type.nullable = true; c.validity = pack_bits(*valid); }
    return c;
  }
  // strings of at most 12 bytes (inline views, binview/view.rs:30-42)
  static Column from_short_strings(const std::vector<std::string>& v) {
    std::vector<uint8_t> views(v.size() * 16, 0);
    for (size_t i = 0; i < v.size(); ++i) {
      if (v[i].size() > 12) throw ErrorCode::Unimplemented("from_short_strings: string longer than 12 bytes");
      uint32_t len = (uint32_t)v[i].size();
      memcpy(&views[i * 16], &len, 4);
      memcpy(&views[i * 16 + 4], v[i].data(), len);
    }
    Column c; c.type = DataType::of(DBHIP_T_STRING); c.len = (int64_t)v.size();
    c.data = make_buf(views.size()); c.data->upload(views.data(), views.size());
    return c;
  }
  // strings of any length: inline views up to 12 bytes, {len, prefix, buffer 0, offset} into ONE data buffer beyond (binview/view.rs)
  static Column from_strings(const std::vector<std::string>& v) {
    std::vector<uint8_t> views(v.size() * 16, 0), bytes;
    for (size_t i = 0; i < v.size(); ++i) {
      const uint32_t len = (uint32_t)v[i].size();
      memcpy(&views[i * 16], &len, 4);
      if (len <= 12) { memcpy(&views[i * 16 + 4], v[i].data(), len); continue; }
      const uint32_t off = (uint32_t)bytes.size(), zero = 0;
      memcpy(&views[i * 16 + 4], v[i].data(), 4);
      memcpy(&views[i * 16 + 8], &zero, 4);
      memcpy(&views[i * 16 + 12], &off, 4);
      bytes.insert(bytes.end(), v[i].begin(), v[i].end());
    }
    Column c; c.type = DataType::of(DBHIP_T_STRING); c.len = (int64_t)v.size();
    c.data = make_buf(views.size() + 16); c.data->upload(views.data(), views.size());
    c.str_data = make_buf(bytes.size() + 16);
    if (!bytes.empty()) c.str_data->upload(bytes.data(), bytes.size());
    const void* p = c.str_data->ptr();
    c.str_ptrs = make_buf(sizeof(void*));
    c.str_ptrs->upload(&p, sizeof(void*));
    return c;
  }
  // the values of `n` views at `views_dev` whose long form points into the `nbytes` bytes at `bytes_dev`
  static std::vector<std::string> strings_to_host(const void* views_dev, int64_t n, const void* bytes_dev, int64_t nbytes) {
    std::vector<uint8_t> views((size_t)n * 16), bytes((size_t)nbytes);
    if (n) check(dbhip_memcpy_d2h(views.data(), views_dev, views.size(), nullptr));
    if (nbytes) check(dbhip_memcpy_d2h(bytes.data(), bytes_dev, bytes.size(), nullptr));
    std::vector<std::string> out((size_t)n);
    for (int64_t i = 0; i < n; ++i) {
      uint32_t l, off; memcpy(&l, &views[(size_t)i * 16], 4); memcpy(&off, &views[(size_t)i * 16 + 12], 4);
      out[(size_t)i] = l <= 12 ? std::string((const char*)&views[(size_t)i * 16 + 4], l) : std::string((const char*)&bytes[off], l);
    }
    return out;
  }
  static Buf pack_bits(const std::vector<bool>& v) {
    std::vector<uint8_t> by((v.size() + 63) / 64 * 8 + 8, 0);
    for (size_t i = 0; i < v.size(); ++i) if (v[i]) by[i >> 3] |= (uint8_t)(1u << (i & 7));
    Buf b = make_buf(by.size()); b->upload(by.data(), by.size());
    return b;
  }
  static std::vector<bool> unpack_bits(const Buf& b, int64_t n) {
    std::vector<uint8_t> by((size_t)(n + 7) / 8);
    b->download(by.data(), by.size());
    std::vector<bool> v((size_t)n);
    for (int64_t i = 0; i < n; ++i) v[(size_t)i] = (by[(size_t)(i >> 3)] >> (i & 7)) & 1;
    return v;
  }
  template <typename T>
  std::vector<T> to_vector() const {
    size_t n = (size_t)len * (type.dim > 0 ? (size_t)type.dim : 1);
    std::vector<T> v(n);
    data->download(v.data(), n * sizeof(T));
    return v;
  }
  std::vector<bool> to_bools() const { return unpack_bits(data, len); }
  std::vector<bool> validity_to_host() const { return validity ? unpack_bits(validity, len) : std::vector<bool>((size_t)len, true); }
  std::vector<std::string> to_short_strings() const {
    std::vector<uint8_t> views((size_t)len * 16);
    data->download(views.data(), views.size());
    std::vector<std::string> out((size_t)len);
    for (int64_t i = 0; i < len; ++i) {
      uint32_t l; memcpy(&l, &views[(size_t)i * 16], 4);
      out[(size_t)i] = std::string((const char*)&views[(size_t)i * 16 + 4], l);
    }
    return out;
  }
  dbhip_col c() const {
    dbhip_col o; memset(&o, 0, sizeof(o));
    o.type = type.id; o.is_scalar = 0; o.data = data ? data->ptr() : nullptr;
    o.validity = validity ? (const uint8_t*)validity->ptr() : nullptr;
    o.buffers = str_ptrs ? (const void* const*)str_ptrs->ptr() : nullptr;
    o.n_buffers = str_ptrs ? 1 : 0;
    o.precision = type.precision; o.scale = type.scale;
    return o;
  }
};

// Value<AnyType> (values.rs:122): Scalar or Column
struct Value {
  bool is_scalar = false;
  Scalar scalar;
  Column column;
  Buf scalar_dev;  // device image of the scalar (one element)
  static Value of(Column c) { Value v; v.column = std::move(c); return v; }
  static Value of(Scalar s) {
    Value v; v.is_scalar = true; v.scalar = s;
    uint8_t raw[16] = {0};
    s.store(raw);
    v.scalar_dev = make_buf(16); v.scalar_dev->upload(raw, 16);
    return v;
  }
  const DataType& type() const { return is_scalar ? scalar.type : column.type; }
  dbhip_col c() const {
    if (!is_scalar) return column.c();
    dbhip_col o; memset(&o, 0, sizeof(o));
    o.type = scalar.type.id; o.is_scalar = 1; o.data = scalar_dev->ptr();
    o.precision = scalar.type.precision; o.scale = scalar.type.scale;
    return o;
  }
};

// DataBlock (block.rs:49-59)
struct DataBlock {
  std::vector<Column> columns;
  int64_t num_rows = 0;
  std::shared_ptr<void> meta;  // BlockMetaInfo (e.g. AggregateMeta::Serialized)
  DataBlock() = default;
  DataBlock(std::vector<Column> cols, int64_t n) : columns(std::move(cols)), num_rows(n) {}
  const Column& get_by_offset(size_t i) const { return columns.at(i); }
  size_t num_columns() const { return columns.size(); }
  bool is_empty() const { return num_rows == 0; }
};

// ---- function machinery (function.rs) ----------------------------------------------------------
struct EvalContext {
  int64_t num_rows = 0;
  Buf validity;  // rows whose bit is 0 never raise (function.rs:536-543)
  // errors: (row validity bitmap on the device: 1 = ok, message of the FIRST error set) — function.rs:534-556
  Buf error_bitmap;
  std::string error_msg;
  bool has_errors() const { return (bool)error_bitmap; }
  void set_errors(Buf bitmap, const std::string& msg) {
    if (!error_bitmap) { error_bitmap = std::move(bitmap); error_msg = msg; }
  }
};

struct FunctionSignature {
  std::string name;
  std::vector<DataType> args_type;
  DataType return_type;
};
struct ScalarFunction {
  virtual ~ScalarFunction() = default;
  virtual Value eval(const std::vector<Value>& args, EvalContext& ctx) const = 0;
};
struct Function {
  FunctionSignature signature;
  std::shared_ptr<ScalarFunction> eval;  // FunctionEval::Scalar
};

class FunctionRegistry {
 public:
  void register_function(Function f) { funcs_[f.signature.name].push_back(std::make_shared<Function>(std::move(f))); }
  // candidates in registration order (overload id, function.rs:275-340,444-447); exact physical match,
  // nullable arguments are accepted by the passthrough_nullable wrapper of the same overload
  std::shared_ptr<Function> search(const std::string& name, const std::vector<DataType>& args) const {
    auto it = funcs_.find(name);
    if (it == funcs_.end()) throw ErrorCode::BadArguments("function `" + name + "` does not exist");
    for (auto& f : it->second) {
      if (f->signature.args_type.size() != args.size()) continue;
      bool ok = true;
      for (size_t i = 0; i < args.size(); ++i) ok &= f->signature.args_type[i].same_physical(args[i]);
      if (ok) return f;
    }
    std::string s = "no overload of `" + name + "` for (";
    for (size_t i = 0; i < args.size(); ++i) s += (i ? ", " : "") + args[i].name();
    throw ErrorCode::BadArguments(s + ")");
  }
  static const FunctionRegistry& builtin();
 private:
  std::map<std::string, std::vector<std::shared_ptr<Function>>> funcs_;
};

// ---- expressions (expression.rs) ---------------------------------------------------------------
struct Expr {
  enum Kind { Constant, ColumnRef, Cast, FunctionCall } kind = Constant;
  Scalar scalar;                         // Constant
  size_t id = 0; std::string display;    // ColumnRef
  DataType type;                         // data type of the node
  std::shared_ptr<Function> function;    // FunctionCall
  std::string fname;
  std::vector<Expr> args;                // FunctionCall args / Cast inner
  const DataType& data_type() const { return type; }
  std::string sql_display() const {
    switch (kind) {
      case Constant: return scalar.to_string();
      case ColumnRef: return display;
      case Cast: return "CAST(" + args[0].sql_display() + " AS " + type.name() + ")";
      default: {
        static const std::map<std::string, std::string> infix = {{"plus", "+"}, {"minus", "-"}, {"multiply", "*"}, {"divide", "/"},
            {"eq", "="}, {"noteq", "<>"}, {"lt", "<"}, {"lte", "<="}, {"gt", ">"}, {"gte", ">="}, {"modulo", "%"}, {"div", "DIV"}};
        auto it = infix.find(fname);
        if (it != infix.end() && args.size() == 2) return "(" + args[0].sql_display() + " " + it->second + " " + args[1].sql_display() + ")";
        std::string s = fname + "(";
        for (size_t i = 0; i < args.size(); ++i) s += (i ? ", " : "") + args[i].sql_display();
        return s + ")";
      }
    }
  }
  static Expr constant(Scalar s) { Expr e; e.kind = Constant; e.scalar = s; e.type = s.type; return e; }
  static Expr column_ref(size_t id, DataType t, std::string name) { Expr e; e.kind = ColumnRef; e.id = id; e.type = t; e.display = std::move(name); return e; }
  static Expr cast(Expr inner, DataType dest) { Expr e; e.kind = Cast; e.type = dest; e.args.push_back(std::move(inner)); return e; }
  // type-checks the call against the registry (type_check::check_function)
  static Expr call(const std::string& name, std::vector<Expr> args, const FunctionRegistry& reg = FunctionRegistry::builtin());
};

// ---- small helpers over the C-ABI ---------------------------------------------------------------
inline Buf and_validity(const Buf& a, const Buf& b, int64_t n) {
  if (!a) return b;
  if (!b) return a;
  Buf out = make_buf((size_t)(n + 63) / 64 * 8 + 8);
  check(dbhip_bitmap_binary(0, (const uint8_t*)a->ptr(), (const uint8_t*)b->ptr(), n, (uint8_t*)out->ptr(), nullptr));
  return out;
}
// op: 1 = a | b, 2 = a & ~b (both operands present)
inline Buf bitmap_op(int op, const Buf& a, const Buf& b, int64_t n) {
  Buf out = make_buf((size_t)(n + 63) / 64 * 8 + 8);
  check(dbhip_bitmap_binary(op, (const uint8_t*)a->ptr(), (const uint8_t*)b->ptr(), n, (uint8_t*)out->ptr(), nullptr));
  return out;
}
inline Buf const_bitmap(bool v, int64_t n) {
  Buf b = make_buf((size_t)(n + 63) / 64 * 8 + 8);
  b->fill(v ? 0xFF : 0x00);
  return b;
}
inline int64_t count_bits(const Buf& bm, int64_t n) {
  Buf c = make_buf(8); c->fill(0);
  check(dbhip_bitmap_count((const uint8_t*)bm->ptr(), 0, n, (uint64_t*)c->ptr(), nullptr));
  uint64_t h = 0; c->download(&h, 8);
  return (int64_t)h;
}
// validity of a Value for `n` rows (NULL scalar -> all-zero bitmap)
inline Buf value_validity(const Value& v, int64_t n) {
  if (v.is_scalar) return v.scalar.is_null ? const_bitmap(false, n) : Buf();
  return v.column.validity;
}

// ---- Evaluator (evaluator.rs:229-464) ----------------------------------------------------------
class Evaluator {
 public:
  Evaluator(const DataBlock& block, const FunctionRegistry& reg = FunctionRegistry::builtin()) : block_(block), reg_(reg) {}
  // Runs `expr` over the block; a Value::Column result always has block.num_rows rows (evaluator.rs:334-349).
  Value run(const Expr& expr) const { return partial_run(expr, Buf(), nullptr); }
  // `selection`: rows a row error is reported for (render_error honours it, function.rs:567-620)
  Value run_with_selection(const Expr& expr, const std::vector<uint32_t>* selection) const { return partial_run(expr, Buf(), selection); }

  // The whole tree in ONE launch (dbhip_expr_eval): the tree is flattened post-order into a register program —
  // what a CompoundBlockOperator / BlockOperator::Map fusion (block_operator.rs:42-85) would hand to the device.
  // Returns nullopt when a node is outside the fused subset (decimals, strings, modulo, checked casts, more than 8
  // registers / input columns): the caller then uses run(), one kernel per node. Row errors of `/` are not rendered
  // here (a failing fused program is re-run node by node to name the failing call like render_error does).
  std::optional<Value> run_fused(const Expr& expr) const {
    std::vector<dbhip_expr_ins> prog;
    std::vector<size_t> inputs;           // block column ids, in input order
    std::vector<int> free_regs = {7, 6, 5, 4, 3, 2, 1, 0};
    bool ok = true;
    std::function<int(const Expr&)> emit = [&](const Expr& e) -> int {
      if (!ok) return 0;
      auto alloc = [&]() -> int { if (free_regs.empty()) { ok = false; return 0; } int r = free_regs.back(); free_regs.pop_back(); return r; };
      auto ins = [&](int op, int dst, int a, int b, int type, uint64_t imm) { dbhip_expr_ins i; memset(&i, 0, sizeof(i)); i.op = op; i.dst = dst; i.a = a; i.b = b; i.type = type; i.imm = imm; prog.push_back(i); };
      const int t = e.type.id;
      const bool plain = (t >= DBHIP_T_BOOL && t <= DBHIP_T_TIMESTAMP) && e.type.dim == 0;
      if (!plain) { ok = false; return 0; }
      switch (e.kind) {
        case Expr::ColumnRef: {
          size_t k = 0;
          for (; k < inputs.size(); ++k) if (inputs[k] == e.id) break;
          if (k == inputs.size()) { if (inputs.size() == 8) { ok = false; return 0; } inputs.push_back(e.id); }
          int r = alloc();
          ins(DBHIP_EX_LOAD, r, (int)k, 0, t, 0);
          return r;
        }
        case Expr::Constant: {
          if (e.scalar.is_null) { ok = false; return 0; }
          uint64_t imm;
          if (t == DBHIP_T_F32) { double d = (double)(float)e.scalar.f; memcpy(&imm, &d, 8); }
          else if (t == DBHIP_T_F64) memcpy(&imm, &e.scalar.f, 8);
          else imm = (uint64_t)e.scalar.i;
          int r = alloc();
          ins(DBHIP_EX_CONST, r, 0, 0, t, imm);
          return r;
        }
        case Expr::Cast: {
          int a = emit(e.args[0]);
          if (!ok) return 0;
          if (e.args[0].type.same_physical(e.type)) return a;
          ins(DBHIP_EX_CAST, a, a, 0, t, 0);
          return a;
        }
        default: break;
      }
      static const std::map<std::string, int> ops = {{"plus", DBHIP_EX_PLUS}, {"minus", DBHIP_EX_MINUS}, {"multiply", DBHIP_EX_MULTIPLY},
          {"divide", DBHIP_EX_DIVIDE}, {"eq", DBHIP_EX_EQ}, {"noteq", DBHIP_EX_NOTEQ}, {"lt", DBHIP_EX_LT}, {"lte", DBHIP_EX_LTE},
          {"gt", DBHIP_EX_GT}, {"gte", DBHIP_EX_GTE}};
      if (e.fname == "and_filters" || e.fname == "or_filters") {   // AND / OR over decoded (NULL -> FALSE) arguments
        int acc = -1;
        for (const Expr& a : e.args) {
          int r = emit(a);
          if (!ok) return 0;
          ins(DBHIP_EX_IS_TRUE, r, r, 0, DBHIP_T_BOOL, 0);
          if (acc < 0) { acc = r; continue; }
          ins(e.fname == "or_filters" ? DBHIP_EX_OR : DBHIP_EX_AND, acc, acc, r, DBHIP_T_BOOL, 0);
          free_regs.push_back(r);
        }
        return acc;
      }
      auto it = ops.find(e.fname);
      if (it == ops.end() || e.args.size() != 2) { ok = false; return 0; }
      int a = emit(e.args[0]);
      int b = emit(e.args[1]);
      if (!ok) return 0;
      ins(it->second, a, a, b, t, 0);
      free_regs.push_back(b);
      return a;
    };
    const int out = emit(expr);
    if (!ok || prog.empty()) return std::nullopt;
    const int64_t n = block_.num_rows;
    std::vector<dbhip_col> cols;
    bool nullable = false;
    for (size_t id : inputs) { cols.push_back(block_.get_by_offset(id).c()); nullable |= (bool)block_.get_by_offset(id).validity; }
    Column c; c.type = expr.type.remove_nullable(); c.len = n;
    const size_t words = (size_t)(n + 63) / 64;
    c.data = make_buf(c.type.id == DBHIP_T_BOOL ? words * 8 + 8 : (size_t)n * c.type.elem_size() + 16);
    if (nullable) { c.validity = make_buf(words * 8 + 8); c.type.nullable = true; }
    Buf err = make_buf((size_t)(n + 31) / 32 * 4 + 8);
    Buf cnt = make_buf(8); cnt->fill(0);
    int32_t rc = dbhip_expr_eval(prog.data(), (int32_t)prog.size(), cols.data(), (int32_t)cols.size(), n, out, c.data->ptr(),
                                 c.validity ? (uint8_t*)c.validity->ptr() : nullptr, (uint8_t*)err->ptr(), (uint64_t*)cnt->ptr(),
                                 nullptr, nullptr);
    if (rc == DBHIP_ERR_UNSUPPORTED || rc == DBHIP_ERR_INVALID) return std::nullopt;
    check(rc);
    uint64_t nerr = 0; cnt->download(&nerr, 8);
    if (nerr) return run(expr);  // names the failing call and row exactly like the reference
    return Value::of(c);
  }

 private:
  Value partial_run(const Expr& e, Buf validity, const std::vector<uint32_t>* selection) const {
    switch (e.kind) {
      case Expr::Constant: return Value::of(e.scalar);
      case Expr::ColumnRef: return Value::of(block_.get_by_offset(e.id));
      case Expr::Cast: return run_cast(e, partial_run(e.args[0], validity, selection));
      default: break;
    }
    if (e.fname == "and_filters") return eval_and_filters(e, validity, selection);
    if (e.fname == "or_filters") return eval_or_filters(e, validity, selection);
    return eval_common_call(e, validity, selection);
  }

  // numeric widening casts ride on the `plus` kernel: x + 0::dest is `x as dest` whenever
  // ResultTypeOfBinary(src, dest) == dest (numeric_basic_arithmetic.rs:255-292 casts both sides first)
  Value run_cast(const Expr& e, Value v) const {
    const DataType& dest = e.type;
    if (v.type().same_physical(dest)) return v;
    if (!(v.type().is_numeric() && dest.is_numeric()) || dbhip_arith_result_type(DBHIP_OP_PLUS, v.type().id, dest.id) != dest.id)
      throw ErrorCode::Unimplemented("CAST(" + v.type().name() + " AS " + dest.name() + ") stays on the CPU evaluator");
    Scalar zero = (dest.id == DBHIP_T_F32 || dest.id == DBHIP_T_F64) ? Scalar::Float(dest.id, 0.0) : Scalar::Int(dest.id, 0);
    std::vector<Value> args = {v, Value::of(zero)};
    EvalContext ctx; ctx.num_rows = block_.num_rows;
    return reg_.search("plus", {v.type(), dest})->eval->eval(args, ctx);
  }

  // evaluator.rs:284-305: conjunction of boolean filters; later predicates only matter where earlier held
  Value eval_and_filters(const Expr& e, Buf validity, const std::vector<uint32_t>* selection) const {
    const int64_t n = block_.num_rows;
    Buf acc;
    for (const Expr& a : e.args) {
      Value v = partial_run(a, validity, selection);
      Buf bits;
      if (v.is_scalar) bits = const_bitmap(!v.scalar.is_null && v.scalar.i != 0, n);
      else bits = and_validity(v.column.data, v.column.validity, n);  // NULL -> false
      acc = and_validity(acc, bits, n);
    }
    Column c; c.type = DataType::of(DBHIP_T_BOOL); c.len = n; c.data = acc ? acc : const_bitmap(true, n);
    return Value::of(c);
  }

  // evaluator.rs:1802-1880: disjunction of decoded predicates (NULL -> false); a later argument is only evaluated (and may only
  // raise) on the rows no earlier argument made TRUE
  Value eval_or_filters(const Expr& e, Buf validity, const std::vector<uint32_t>* selection) const {
    const int64_t n = block_.num_rows;
    Buf result;
    for (const Expr& a : e.args) {
      Value v = partial_run(a, validity, selection);
      Buf bits;
      if (v.is_scalar) bits = const_bitmap(!v.scalar.is_null && v.scalar.i != 0, n);
      else bits = and_validity(v.column.data, v.column.validity, n);  // NULL -> false
      result = result ? bitmap_op(1, result, bits, n) : bits;
      validity = bitmap_op(2, validity ? validity : const_bitmap(true, n), bits, n);
    }
    Column c; c.type = DataType::of(DBHIP_T_BOOL); c.len = n; c.data = result ? result : const_bitmap(false, n);
    return Value::of(c);
  }

  Value eval_common_call(const Expr& e, Buf validity, const std::vector<uint32_t>* selection) const {
    std::vector<Value> args;
    for (const Expr& a : e.args) args.push_back(partial_run(a, validity, selection));
    bool all_scalar = true;
    for (const Value& v : args) {
      all_scalar &= v.is_scalar;
      if (!v.is_scalar && !v.column.is_const && v.column.len != block_.num_rows) throw ErrorCode::Internal("argument column length != num_rows");
    }
    EvalContext ctx; ctx.num_rows = block_.num_rows;
    ctx.validity = all_scalar ? Buf() : validity;
    Value result = e.function->eval->eval(args, ctx);
    render_error(e, ctx, args, selection);
    if (!result.is_scalar && result.column.len != block_.num_rows) throw ErrorCode::Internal("result length != num_rows");
    return result;
  }

  // EvalContext::render_error (function.rs:567-620): first failing row (honouring the selection) and
  // "<msg> while evaluating function `f(args)` in expr `...`"
  void render_error(const Expr& e, const EvalContext& ctx, const std::vector<Value>& args, const std::vector<uint32_t>* selection) const {
    if (!ctx.has_errors()) return;
    std::vector<bool> ok = Column::unpack_bits(ctx.error_bitmap, ctx.num_rows);
    int64_t first = -1;
    if (!selection) { for (int64_t i = 0; i < ctx.num_rows && first < 0; ++i) if (!ok[(size_t)i]) first = i; }
    else { for (uint32_t r : *selection) if (!ok[r]) { first = r; break; } }
    if (first < 0) return;
    std::string a;
    for (size_t k = 0; k < args.size(); ++k) a += (k ? ", " : "") + value_at(args[k], first);
    throw ErrorCode::BadArguments(ctx.error_msg + " while evaluating function `" + e.fname + "(" + a + ")` in expr `" + e.sql_display() + "`");
  }
  static std::string value_at(const Value& v, int64_t row) {
    if (v.is_scalar) return v.scalar.to_string();
    const Column& c = v.column;
    if (c.validity && !Column::unpack_bits(c.validity, c.len)[(size_t)row]) return "NULL";
    uint8_t raw[16] = {0};
    size_t es = c.type.elem_size();
    if (es == 0 || es > 16) return "?";
    std::vector<uint8_t> all((size_t)c.len * es);
    c.data->download(all.data(), all.size());
    memcpy(raw, &all[(size_t)row * es], es);
    switch (c.type.id) {
      case DBHIP_T_I8: return std::to_string(*(int8_t*)raw);
      case DBHIP_T_I16: return std::to_string(*(int16_t*)raw);
      case DBHIP_T_I32: case DBHIP_T_DATE: return std::to_string(*(int32_t*)raw);
      case DBHIP_T_U8: return std::to_string(*(uint8_t*)raw);
      case DBHIP_T_U16: return std::to_string(*(uint16_t*)raw);
      case DBHIP_T_U32: return std::to_string(*(uint32_t*)raw);
      case DBHIP_T_U64: return std::to_string(*(uint64_t*)raw);
      case DBHIP_T_F32: { std::ostringstream o; o << *(float*)raw; return o.str(); }
      case DBHIP_T_F64: { std::ostringstream o; o << *(double*)raw; return o.str(); }
      default: return std::to_string(*(int64_t*)raw);
    }
  }

  const DataBlock& block_;
  const FunctionRegistry& reg_;
};

// ---- built-in functions: thin adapters from ScalarFunction::eval to the C-ABI -------------------
namespace detail {

inline int64_t rows_of(const std::vector<Value>& args, const EvalContext& ctx) {
  for (const Value& v : args) if (!v.is_scalar) return v.column.len;
  return ctx.num_rows > 0 ? 1 : 0;  // all-scalar call: evaluated once (evaluator.rs:413-420)
}

// passthrough_nullable (register_vectorize.rs:447-471): validity = AND of the inputs
inline Buf merged_validity(const std::vector<Value>& args, int64_t n) {
  Buf acc;
  for (const Value& v : args) acc = and_validity(acc, value_validity(v, n), n);
  return acc;
}

struct ArithFn : ScalarFunction {
  int op; DataType out;
  ArithFn(int op_, DataType o) : op(op_), out(o) {}
  Value eval(const std::vector<Value>& args, EvalContext& ctx) const override {
    const int64_t n = rows_of(args, ctx);
    Column r; r.type = out; r.len = n; r.data = make_buf((size_t)n * out.elem_size());
    r.validity = merged_validity(args, n);
    r.type.nullable = (bool)r.validity;
    dbhip_col a = args[0].c(), b = args[1].c();
    // rows that are NULL (or masked by ctx.validity) never raise: pass the effective validity
    Buf eff = and_validity(r.validity, ctx.validity, n);
    if (eff) { if (!a.is_scalar) a.validity = (const uint8_t*)eff->ptr(); else if (!b.is_scalar) b.validity = (const uint8_t*)eff->ptr(); }
    const bool can_fail = op >= DBHIP_OP_DIVIDE;
    Buf err = can_fail ? const_bitmap(true, n) : Buf();
    Buf cnt = can_fail ? make_buf(8) : Buf();
    if (cnt) cnt->fill(0);
    int32_t rc = dbhip_arith(op, &a, &b, n, out.id, r.data->ptr(), err ? (uint8_t*)err->ptr() : nullptr,
                             cnt ? (uint64_t*)cnt->ptr() : nullptr, nullptr);
    check(rc);
    if (op == DBHIP_OP_DIVNULL) {  // x / 0 = NULL: the kernel's row bitmap is the result's validity contribution
      uint64_t nnull = 0; cnt->download(&nnull, 8);
      if (nnull) { r.validity = and_validity(r.validity, err, n); r.type.nullable = true; }
    } else if (can_fail && op != DBHIP_OP_DIV0) {  // EvalContext::set_error replay: the kernel cleared the bit of every failing row
      uint64_t nerr = 0; cnt->download(&nerr, 8);
      if (nerr) ctx.set_errors(err, op == DBHIP_OP_MODULO ? "Division by zero" : "divided by zero");
    }
    if (args[0].is_scalar && args[1].is_scalar) return scalar_of(r);
    return Value::of(r);
  }
  static Value scalar_of(const Column& c) {  // all-scalar call -> Scalar result
    Scalar s; s.type = c.type;
    if (c.validity && !Column::unpack_bits(c.validity, 1)[0]) { s.is_null = true; return Value::of(s); }
    uint8_t raw[16] = {0};
    c.data->download(raw, c.type.elem_size());
    switch (c.type.id) {
      case DBHIP_T_F32: s.f = *(float*)raw; break;
      case DBHIP_T_F64: s.f = *(double*)raw; break;
      case DBHIP_T_DEC128: memcpy(&s.dec, raw, 16); break;
      case DBHIP_T_I8: s.i = *(int8_t*)raw; break;
      case DBHIP_T_I16: s.i = *(int16_t*)raw; break;
      case DBHIP_T_I32: case DBHIP_T_DATE: s.i = *(int32_t*)raw; break;
      case DBHIP_T_U8: s.i = *(uint8_t*)raw; break;
      case DBHIP_T_U16: s.i = *(uint16_t*)raw; break;
      case DBHIP_T_U32: s.i = *(uint32_t*)raw; break;
      default: memcpy(&s.i, raw, 8); break;
    }
    s.u = (uint64_t)s.i; if (c.type.id == DBHIP_T_DEC64) s.dec = s.i;
    return Value::of(s);
  }
};

struct DecimalFn : ScalarFunction {
  int op; DataType out;
  DecimalFn(int op_, DataType o) : op(op_), out(o) {}
  Value eval(const std::vector<Value>& args, EvalContext& ctx) const override {
    const int64_t n = rows_of(args, ctx);
    Column r; r.type = out; r.len = n; r.data = make_buf((size_t)n * out.elem_size());
    r.validity = merged_validity(args, n);
    r.type.nullable = (bool)r.validity;
    dbhip_col a = args[0].c(), b = args[1].c();
    Buf err = const_bitmap(true, n);
    Buf cnt = make_buf(8); cnt->fill(0);
    int32_t rc = dbhip_decimal_arith(op, &a, &b, n, out.id, out.precision, out.scale, r.data->ptr(), (uint8_t*)err->ptr(),
                                     (uint64_t*)cnt->ptr(), nullptr);
    check(rc);
    uint64_t nerr = 0; cnt->download(&nerr, 8);
    if (nerr) ctx.set_errors(err, op == DBHIP_OP_DIVIDE ? "divided by zero" : "Decimal overflow");
    return Value::of(r);
  }
};

struct CmpFn : ScalarFunction {
  int op;
  explicit CmpFn(int op_) : op(op_) {}
  Value eval(const std::vector<Value>& args, EvalContext& ctx) const override {
    const int64_t n = rows_of(args, ctx);
    Column r; r.type = DataType::of(DBHIP_T_BOOL); r.len = n; r.data = make_buf((size_t)(n + 63) / 64 * 8 + 8);
    r.validity = merged_validity(args, n);
    r.type.nullable = (bool)r.validity;
    dbhip_col a = args[0].c(), b = args[1].c();
    check(dbhip_cmp(op, &a, &b, n, (uint8_t*)r.data->ptr(), nullptr));
    return Value::of(r);
  }
};

struct VecDistFn : ScalarFunction {
  int metric;
  explicit VecDistFn(int m) : metric(m) {}
  // cosine_distance(Vector(N) column, Vector(N) scalar/column of ONE query): Float32 per row
  // (scalars/vector.rs:497-560 evaluates the pair row by row; the query side is a constant in
  // the ORDER BY distance plans this serves)
  Value eval(const std::vector<Value>& args, EvalContext&) const override {
    const Column& base = args[0].column;
    const Column& q = args[1].column;
    if (!q.is_const && q.len != 1) {
      // two columns: row by row (calculate_distance, scalars/vector.rs:490-560)
      if (q.len != base.len || q.type.dim != base.type.dim) throw ErrorCode::BadArguments("Vector length not equal");
      Column r; r.type = DataType::of(DBHIP_T_F32); r.len = base.len; r.data = make_buf((size_t)base.len * 4 + 16);
      check(dbhip_vec_distance_rows(metric, DBHIP_T_F32, base.data->ptr(), 0, q.data->ptr(), 0, base.len, base.type.dim, r.data->ptr(), nullptr));
      return Value::of(r);
    }
    Column r; r.type = DataType::of(DBHIP_T_F32); r.len = base.len; r.data = make_buf((size_t)base.len * 4);
    check(dbhip_vec_distance(metric, (const float*)base.data->ptr(), base.len, base.type.dim, (const float*)q.data->ptr(), 1,
                             (float*)r.data->ptr(), nullptr));
    return Value::of(r);
  }
};

inline void register_builtins(FunctionRegistry& reg) {
  static const int num_types[] = {DBHIP_T_I8, DBHIP_T_I16, DBHIP_T_I32, DBHIP_T_I64, DBHIP_T_U8, DBHIP_T_U16, DBHIP_T_U32, DBHIP_T_U64,
                                  DBHIP_T_F32, DBHIP_T_F64};
  static const std::pair<const char*, int> ops[] = {{"plus", DBHIP_OP_PLUS}, {"minus", DBHIP_OP_MINUS}, {"multiply", DBHIP_OP_MULTIPLY},
                                                    {"divide", DBHIP_OP_DIVIDE}, {"div", DBHIP_OP_INTDIV}, {"modulo", DBHIP_OP_MODULO}};
  static const std::pair<const char*, int> cmps[] = {{"eq", DBHIP_CMP_EQ}, {"noteq", DBHIP_CMP_NOTEQ}, {"lt", DBHIP_CMP_LT},
                                                     {"lte", DBHIP_CMP_LTE}, {"gt", DBHIP_CMP_GT}, {"gte", DBHIP_CMP_GTE}};
  // register_basic_arithmetic (integer_arithmetic.rs:26-45, numeric_basic_arithmetic.rs:546-605): every numeric pair
  for (auto& op : ops)
    for (int l : num_types)
      for (int r : num_types) {
        int out = dbhip_arith_result_type(op.second, l, r);
        if (out < 0) continue;
        reg.register_function({{op.first, {DataType::of(l), DataType::of(r)}, DataType::of(out)},
                               std::make_shared<ArithFn>(op.second, DataType::of(out))});
      }
  // div0 / divnull are registered on Float64 only (register_div_arithmetic, numeric_basic_arithmetic.rs:524-543)
  reg.register_function({{"div0", {DataType::of(DBHIP_T_F64), DataType::of(DBHIP_T_F64)}, DataType::of(DBHIP_T_F64)},
                         std::make_shared<ArithFn>(DBHIP_OP_DIV0, DataType::of(DBHIP_T_F64))});
  reg.register_function({{"divnull", {DataType::of(DBHIP_T_F64), DataType::of(DBHIP_T_F64)}, DataType::of(DBHIP_T_F64, true)},
                         std::make_shared<ArithFn>(DBHIP_OP_DIVNULL, DataType::of(DBHIP_T_F64))});
  // comparisons: same physical type on both sides (the planner inserts casts, comparison.rs:98-112)
  static const int cmp_types[] = {DBHIP_T_BOOL, DBHIP_T_I8, DBHIP_T_I16, DBHIP_T_I32, DBHIP_T_I64, DBHIP_T_U8, DBHIP_T_U16, DBHIP_T_U32,
                                  DBHIP_T_U64, DBHIP_T_F32, DBHIP_T_F64, DBHIP_T_DATE, DBHIP_T_TIMESTAMP, DBHIP_T_STRING};
  for (auto& c : cmps)
    for (int t : cmp_types)
      reg.register_function({{c.first, {DataType::of(t), DataType::of(t)}, DataType::of(DBHIP_T_BOOL)}, std::make_shared<CmpFn>(c.second)});
}

}  // namespace detail

inline const FunctionRegistry& FunctionRegistry::builtin() {
  static FunctionRegistry* reg = [] { auto* r = new FunctionRegistry(); detail::register_builtins(*r); return r; }();
  return *reg;
}

// Expr::call: resolves the overload. Decimal arithmetic / decimal comparisons / vector distances are
// "factory" functions in the reference (resolved per argument types: decimal/src/arithmetic.rs:80-139,
// scalars/vector.rs:262-341) and are built here the same way.
inline Expr Expr::call(const std::string& name, std::vector<Expr> args, const FunctionRegistry& reg) {
  Expr e; e.kind = FunctionCall; e.fname = name;
  std::vector<DataType> at;
  for (auto& a : args) at.push_back(a.data_type());
  e.args = std::move(args);
  static const std::map<std::string, int> arith = {{"plus", DBHIP_OP_PLUS}, {"minus", DBHIP_OP_MINUS}, {"multiply", DBHIP_OP_MULTIPLY}, {"divide", DBHIP_OP_DIVIDE}};
  static const std::map<std::string, int> cmps = {{"eq", DBHIP_CMP_EQ}, {"noteq", DBHIP_CMP_NOTEQ}, {"lt", DBHIP_CMP_LT}, {"lte", DBHIP_CMP_LTE}, {"gt", DBHIP_CMP_GT}, {"gte", DBHIP_CMP_GTE}};
  static const std::map<std::string, int> vec = {{"cosine_distance", DBHIP_VEC_COSINE}, {"l2_distance", DBHIP_VEC_L2}, {"inner_product", DBHIP_VEC_DOT}, {"l1_distance", DBHIP_VEC_L1}};
  if (name == "and_filters" || name == "or_filters") { e.type = DataType::of(DBHIP_T_BOOL); return e; }
  const bool any_decimal = at.size() == 2 && (at[0].is_decimal() || at[1].is_decimal());
  if (any_decimal && arith.count(name)) {
    auto props = [](const DataType& t, uint8_t& p, uint8_t& s) {  // integer -> decimal size (cast.rs:701-753)
      if (t.is_decimal()) { p = t.precision; s = t.scale; return; }
      static const std::map<int, int> ip = {{DBHIP_T_I8, 3}, {DBHIP_T_U8, 3}, {DBHIP_T_I16, 5}, {DBHIP_T_U16, 5}, {DBHIP_T_I32, 10}, {DBHIP_T_U32, 10}, {DBHIP_T_I64, 19}, {DBHIP_T_U64, 20}};
      auto it = ip.find(t.id);
      if (it == ip.end()) throw ErrorCode::BadArguments("decimal arithmetic with " + t.name());
      p = (uint8_t)it->second; s = 0;
    };
    uint8_t lp, ls, rp, rs, op_, os_;
    props(at[0], lp, ls); props(at[1], rp, rs);
    check(dbhip_decimal_result_size(arith.at(name), lp, ls, rp, rs, &op_, &os_));
    e.type = DataType::Decimal(op_, os_);
    e.function = std::make_shared<Function>(Function{{name, at, e.type}, std::make_shared<detail::DecimalFn>(arith.at(name), e.type)});
    return e;
  }
  if (any_decimal && cmps.count(name)) {
    if (!at[0].same_physical(at[1])) throw ErrorCode::BadArguments("decimal comparison needs equal (precision, scale): cast first");
    e.type = DataType::of(DBHIP_T_BOOL);
    e.function = std::make_shared<Function>(Function{{name, at, e.type}, std::make_shared<detail::CmpFn>(cmps.at(name))});
    return e;
  }
  if (vec.count(name)) {
    if (at.size() != 2 || at[0].dim <= 0 || at[0].dim != at[1].dim) throw ErrorCode::BadArguments(name + " needs two Vector(N) arguments of equal N");
    e.type = DataType::of(DBHIP_T_F32);
    e.function = std::make_shared<Function>(Function{{name, at, e.type}, std::make_shared<detail::VecDistFn>(vec.at(name))});
    return e;
  }
  e.function = reg.search(name, at);
  e.type = e.function->signature.return_type;
  for (auto& t : at) if (t.nullable) e.type.nullable = true;
  return e;
}

// ---- FilterExecutor (filter/filter_executor.rs:81-118) + DataBlock::take -------------------------
struct Selection { Buf sel; int64_t count = 0; };

inline Selection filter_select(const Column& predicate) {
  if (predicate.type.id != DBHIP_T_BOOL) throw ErrorCode::BadArguments("filter predicate must be Boolean");
  Buf bits = and_validity(predicate.data, predicate.validity, predicate.len);  // NULL -> not selected
  Selection s; s.sel = make_buf((size_t)predicate.len * 4 + 64);
  Buf cnt = make_buf(8);
  check(dbhip_filter_select((const uint8_t*)bits->ptr(), 0, predicate.len, (uint32_t*)s.sel->ptr(), (uint64_t*)cnt->ptr(), nullptr));
  uint64_t h = 0; cnt->download(&h, 8);
  s.count = (int64_t)h;
  return s;
}

inline Column take(const Column& c, const uint32_t* selp, int64_t k) {
  Column r; r.type = c.type; r.len = k; r.str_data = c.str_data; r.str_ptrs = c.str_ptrs;
  if (c.type.id == DBHIP_T_BOOL) {
    r.data = make_buf((size_t)(k + 63) / 64 * 8 + 8);
    check(dbhip_take_bitmap((const uint8_t*)c.data->ptr(), 0, selp, k, (uint8_t*)r.data->ptr(), nullptr));
  } else {
    size_t es = c.type.elem_size();
    if (es != 1 && es != 2 && es != 4 && es != 8 && es != 16) throw ErrorCode::Unimplemented("take on " + c.type.name());
    r.data = make_buf((size_t)k * es);
    check(dbhip_take(c.data->ptr(), (int32_t)es, selp, k, r.data->ptr(), nullptr));
  }
  if (c.validity) {
    r.validity = make_buf((size_t)(k + 63) / 64 * 8 + 8);
    check(dbhip_take_bitmap((const uint8_t*)c.validity->ptr(), 0, selp, k, (uint8_t*)r.validity->ptr(), nullptr));
  }
  return r;
}

inline Column take(const Column& c, const Buf& sel, int64_t k) { return take(c, (const uint32_t*)sel->ptr(), k); }

inline DataBlock take_block(const DataBlock& b, const uint32_t* sel, int64_t k) {
  DataBlock o; o.num_rows = k;
  for (const Column& c : b.columns) o.columns.push_back(take(c, sel, k));
  return o;
}
inline DataBlock take_block(const DataBlock& b, const Buf& sel, int64_t k) { return take_block(b, (const uint32_t*)sel->ptr(), k); }

// Selector (filter/selector.rs:64-330, select_expr.rs:40-120): the filter's predicate as a tree of And / Or over comparison leaves
// (column-or-constant operands of one fixed-width type) walks TRUE / FALSE lists of row ids instead of materialising one Boolean
// column per node: a conjunct is evaluated only on the rows that passed the conjuncts before it, a disjunct only on the rows that
// failed the disjuncts before it (dbhip_select_cmp). Anything else in the tree (arithmetic under a comparison, casts, strings)
// makes build() return false and the caller keeps the Bitmap path (FilterExecutor::filter) — SelectExpr::Others in the reference.
class Selector {
 public:
  explicit Selector(const DataBlock& block) : block_(block) {}
  // -> the rows that pass, in the reference's true_selection order; nullopt = the tree is outside the leaf kernels
  std::optional<Selection> select(const Expr& e) const {
    if (!supported(e)) return std::nullopt;
    Lists r = run(e, Buf(), block_.num_rows, false);
    Selection s; s.sel = r.t; s.count = r.nt;
    if (!s.sel) { s.sel = make_buf(64); }
    return s;
  }
 private:
  struct Lists { Buf t; int64_t nt = 0; Buf f; int64_t nf = 0; };
  static bool is_cmp(const std::string& n) { return n == "eq" || n == "noteq" || n == "lt" || n == "lte" || n == "gt" || n == "gte"; }
  static int cmp_code(const std::string& n) {
    return n == "eq" ? DBHIP_CMP_EQ : n == "noteq" ? DBHIP_CMP_NOTEQ : n == "lt" ? DBHIP_CMP_LT : n == "lte" ? DBHIP_CMP_LTE : n == "gt" ? DBHIP_CMP_GT : DBHIP_CMP_GTE;
  }
  static bool leaf_operand(const Expr& e) { return e.kind == Expr::ColumnRef || (e.kind == Expr::Constant && !e.scalar.is_null); }
  bool supported(const Expr& e) const {
    if (e.kind == Expr::ColumnRef) return e.type.id == DBHIP_T_BOOL;
    if (e.kind != Expr::FunctionCall) return false;
    if (e.fname == "and_filters" || e.fname == "or_filters" || e.fname == "and" || e.fname == "or") {
      for (const Expr& a : e.args) if (!supported(a)) return false;
      return !e.args.empty();
    }
    if (!is_cmp(e.fname) || e.args.size() != 2 || !leaf_operand(e.args[0]) || !leaf_operand(e.args[1])) return false;
    const DataType &a = e.args[0].type, &b = e.args[1].type;
    const bool fixed = (a.id >= DBHIP_T_BOOL && a.id <= DBHIP_T_TIMESTAMP) || a.id == DBHIP_T_DEC64 || a.id == DBHIP_T_DEC128;
    return fixed && a.dim == 0 && a.id == b.id && a.scale == b.scale && !(e.args[0].kind == Expr::Constant && e.args[1].kind == Expr::Constant);
  }
  dbhip_col operand(const Expr& e, std::vector<Column>& keep) const {
    if (e.kind == Expr::ColumnRef) return block_.get_by_offset(e.id).c();
    keep.push_back(Column::from_scalar(e.scalar));
    dbhip_col c = keep.back().c();
    c.is_scalar = 1;
    return c;
  }
  Lists leaf(const Expr& e, const Buf& sel, int64_t n, bool want_false) const {
    Lists r;
    r.t = make_buf((size_t)(n > 0 ? n : 1) * 4 + 64);
    if (want_false) r.f = make_buf((size_t)(n > 0 ? n : 1) * 4 + 64);
    Buf cnt = make_buf(8);
    const uint32_t* sp = sel ? (const uint32_t*)sel->ptr() : nullptr;
    if (e.kind == Expr::ColumnRef) {
      dbhip_col p = block_.get_by_offset(e.id).c();
      check(dbhip_select_bool(&p, sp, n, (uint32_t*)r.t->ptr(), r.f ? (uint32_t*)r.f->ptr() : nullptr, (uint64_t*)cnt->ptr(), nullptr));
    } else {
      std::vector<Column> keep;
      keep.reserve(2);
      dbhip_col a = operand(e.args[0], keep), b = operand(e.args[1], keep);
      check(dbhip_select_cmp(cmp_code(e.fname), &a, &b, sp, n, (uint32_t*)r.t->ptr(), r.f ? (uint32_t*)r.f->ptr() : nullptr, (uint64_t*)cnt->ptr(), nullptr));
    }
    uint64_t h = 0; cnt->download(&h, 8);
    r.nt = (int64_t)h; r.nf = n - r.nt;
    return r;
  }
  static Buf concat(const std::vector<std::pair<Buf, int64_t>>& parts, int64_t* total) {
    int64_t n = 0;
    for (auto& p : parts) n += p.second;
    Buf out = make_buf((size_t)(n > 0 ? n : 1) * 4 + 64);
    int64_t off = 0;
    for (auto& p : parts) {
      if (p.second) check(dbhip_memcpy_d2d((uint32_t*)out->ptr() + off, p.first->ptr(), (size_t)p.second * 4, nullptr));
      off += p.second;
    }
    *total = n;
    return out;
  }
  // process_select_expr: `sel` (nullptr = all rows) holds the n rows still to be decided
  Lists run(const Expr& e, Buf sel, int64_t n, bool want_false) const {
    const bool is_and = e.kind == Expr::FunctionCall && (e.fname == "and_filters" || e.fname == "and");
    const bool is_or = e.kind == Expr::FunctionCall && (e.fname == "or_filters" || e.fname == "or");
    if (!is_and && !is_or) return leaf(e, sel, n, want_false);
    Lists out;
    if (is_and) {   // process_and (:181-231): the true list narrows, every conjunct's false rows are false
      Buf cur = sel; int64_t k = n;
      std::vector<std::pair<Buf, int64_t>> fparts;
      for (const Expr& c : e.args) {
        Lists r = run(c, cur, k, want_false);
        if (want_false && r.nf) fparts.push_back({r.f, r.nf});
        cur = r.t; k = r.nt;
        if (k == 0) break;
      }
      out.t = cur; out.nt = k;
      if (want_false) out.f = concat(fparts, &out.nf);
      return out;
    }
    // process_or (:233-292): the false list of a disjunct feeds the next one, the true lists are concatenated
    Buf cur = sel; int64_t k = n;
    std::vector<std::pair<Buf, int64_t>> tparts;
    for (const Expr& c : e.args) {
      Lists r = run(c, cur, k, true);
      if (r.nt) tparts.push_back({r.t, r.nt});
      cur = r.f; k = r.nf;
      if (k == 0) break;
    }
    out.t = concat(tparts, &out.nt);
    if (want_false) { out.f = cur; out.nf = k; }
    return out;
  }
  const DataBlock& block_;
};

class FilterExecutor {
 public:
  explicit FilterExecutor(Expr predicate) : predicate_(std::move(predicate)) {}
  // FilterExecutor::select (filter_executor.rs:81-118). The reference walks true / false lists (Selector) so that a CPU core skips
  // the rows already decided; on the device one pass per predicate over the WHOLE column at the streaming rate plus a Bitmap AND
  // is as fast when the first conjunct keeps 5 % of 600 M rows (3.38 vs 3.49 ms) and 2.4x faster when it keeps 98 % (3.46 vs
  // 8.19 ms; tools/bench_selector.py), so the Bitmap path is the default and the Selector — same rows, the reference's list order —
  // is there for a caller that needs the lists themselves (filter_with_selector; falls back for SelectExpr::Others).
  DataBlock filter(const DataBlock& block) const { return filter_with_bitmap(block); }
  DataBlock filter_with_selector(const DataBlock& block) const {
    if (auto s = Selector(block).select(predicate_)) return take_block(block, s->sel, s->count);
    return filter_with_bitmap(block);
  }
  DataBlock filter_with_bitmap(const DataBlock& block) const {
    Evaluator ev(block);
    Value v = ev.run(predicate_);
    if (v.is_scalar) return (!v.scalar.is_null && v.scalar.i != 0) ? block : DataBlock(std::vector<Column>(), 0);
    Selection s = filter_select(v.column);
    return take_block(block, s.sel, s.count);
  }
 private:
  Expr predicate_;
};

// ---- pipeline traits (transform.rs:30-48, transform_accumulating.rs:30-38) ----------------------
struct Transform {
  virtual ~Transform() = default;
  virtual const char* name() const = 0;
  virtual DataBlock transform(DataBlock block) = 0;
  virtual void on_start() {}
  virtual void on_finish() {}
};
struct AccumulatingTransform {
  virtual ~AccumulatingTransform() = default;
  virtual const char* name() const = 0;
  virtual std::vector<DataBlock> transform(DataBlock block) = 0;
  virtual std::vector<DataBlock> on_finish(bool output) = 0;
};

class TransformFilter : public Transform {
 public:
  explicit TransformFilter(Expr predicate) : exec_(std::move(predicate)) {}
  const char* name() const override { return "FilterTransform"; }
  DataBlock transform(DataBlock block) override { return exec_.filter(block); }
 private:
  FilterExecutor exec_;
};

// CompoundBlockOperator::Map (src/query/sql/src/evaluator/block_operator.rs:42-85): appends one column per expr
class TransformMap : public Transform {
 public:
  explicit TransformMap(std::vector<Expr> exprs) : exprs_(std::move(exprs)) {}
  const char* name() const override { return "CompoundBlockOperator"; }
  DataBlock transform(DataBlock block) override {
    for (const Expr& e : exprs_) {
      Evaluator ev(block);
      Value v = ev.run(e);
      if (v.is_scalar) throw ErrorCode::Unimplemented("constant map expression");
      block.columns.push_back(v.column);
    }
    return block;
  }
 private:
  std::vector<Expr> exprs_;
};

// ---- aggregation (aggregate_hashtable.rs, aggregator/*.rs) ---------------------------------------
struct AggregateFunctionDesc {
  std::string name;          // "sum" | "count" | "min" | "max"
  std::optional<size_t> arg; // argument column offset in the input block (count(*) has none)
  DataType arg_type;
};
struct AggregatorParams {  // aggregator_params.rs:16-118
  std::vector<size_t> group_columns;
  std::vector<DataType> group_data_types;
  std::vector<AggregateFunctionDesc> aggregate_functions;
};

struct AggregateMetaSerialized {  // AggregateMeta::Serialized: partial states as device rows
  Buf rows; int64_t n_rows = 0; int64_t row_bytes = 0;
};

class AggregateHashTable {
 public:
  explicit AggregateHashTable(const AggregatorParams& p, int64_t capacity = 1024) : params_(p) {
    std::vector<int32_t> kt; std::vector<uint8_t> kn; std::vector<dbhip_agg_desc> ad;
    for (auto& t : p.group_data_types) { kt.push_back(t.id); kn.push_back(t.nullable ? 1 : 0); }
    for (auto& a : p.aggregate_functions) {
      dbhip_agg_desc d; memset(&d, 0, sizeof(d));
      d.kind = a.name == "count" ? DBHIP_AGG_COUNT : a.name == "sum" ? DBHIP_AGG_SUM : a.name == "min" ? DBHIP_AGG_MIN : a.name == "max" ? DBHIP_AGG_MAX : -1;
      if (d.kind < 0) throw ErrorCode::Unimplemented("aggregate function `" + a.name + "` stays on the CPU operator");
      d.arg_type = a.arg ? a.arg_type.id : 0; d.arg_precision = a.arg_type.precision; d.arg_scale = a.arg_type.scale;
      d.arg_nullable = a.arg && a.arg_type.nullable;
      ad.push_back(d);
    }
    descs_ = ad;
    check(dbhip_groupby_create(kt.data(), kn.data(), (int32_t)kt.size(), ad.data(), (int32_t)ad.size(), capacity, &h_));
  }
  ~AggregateHashTable() { if (h_) dbhip_groupby_destroy(h_); }
  AggregateHashTable(const AggregateHashTable&) = delete;

  // add_groups (aggregate_hashtable.rs:168-207): group columns and aggregate arguments of one block
  void add_groups(const DataBlock& block) {
    std::vector<dbhip_col> keys, args(params_.aggregate_functions.size());
    for (size_t g : params_.group_columns) keys.push_back(block.get_by_offset(g).c());
    for (size_t a = 0; a < args.size(); ++a) {
      memset(&args[a], 0, sizeof(dbhip_col));
      if (params_.aggregate_functions[a].arg) args[a] = block.get_by_offset(*params_.aggregate_functions[a].arg).c();
    }
    check(dbhip_groupby_add_block(h_, keys.data(), args.data(), block.num_rows, nullptr));
  }
  int64_t len() const { int64_t n = 0; check(dbhip_groupby_num_groups(h_, &n, nullptr)); return n; }
  // partial states for the exchange / final stage (Payload::aggregate_flush, payload_flush.rs:151-181)
  AggregateMetaSerialized serialize() const {
    AggregateMetaSerialized m;
    check(dbhip_groupby_row_bytes(h_, &m.row_bytes));
    int64_t g = len();
    m.rows = make_buf((size_t)(g > 0 ? g : 1) * (size_t)m.row_bytes);
    check(dbhip_groupby_flush_serialized(h_, m.rows->ptr(), g, &m.n_rows, nullptr));
    return m;
  }
  // combine_payload (aggregate_hashtable.rs:349-380)
  void combine(const AggregateMetaSerialized& m) { check(dbhip_groupby_merge_serialized(h_, m.rows->ptr(), m.n_rows, nullptr)); }
  // merge_result (:382-408): [aggregate results..., group columns...]
  DataBlock merge_result() const {
    int64_t g = len();
    int64_t cap = g > 0 ? g : 1;
    std::vector<Column> keys, aggs;
    std::vector<void*> kp, ap; std::vector<uint8_t*> kv;
    for (auto& t : params_.group_data_types) {
      Column c; c.type = t; c.len = g; c.data = make_buf((size_t)cap * (t.id == DBHIP_T_BOOL ? 1 : t.elem_size()));
      c.validity = make_buf((size_t)(cap + 31) / 32 * 4 + 8);
      kp.push_back(c.data->ptr()); kv.push_back((uint8_t*)c.validity->ptr());
      keys.push_back(c);
    }
    for (auto& d : descs_) {
      int32_t t; uint8_t p, s;
      check(dbhip_groupby_result_type(&d, &t, &p, &s));
      Column c; c.type = DataType::of(t); c.type.precision = p; c.type.scale = s; c.len = g;
      c.data = make_buf((size_t)cap * c.type.elem_size());
      ap.push_back(c.data->ptr());
      aggs.push_back(c);
    }
    int64_t n = 0;
    check(dbhip_groupby_flush_result(h_, kp.data(), kv.data(), ap.data(), nullptr, cap, &n, nullptr));
    for (size_t k = 0; k < keys.size(); ++k) if (!params_.group_data_types[k].nullable) keys[k].validity.reset();
    DataBlock out; out.num_rows = n;
    for (auto& c : aggs) { c.len = n; out.columns.push_back(c); }
    for (auto& c : keys) { c.len = n; out.columns.push_back(c); }
    return out;
  }
  dbhip_groupby* handle() const { return h_; }
 private:
  AggregatorParams params_;
  std::vector<dbhip_agg_desc> descs_;
  dbhip_groupby* h_ = nullptr;
};

// TransformPartialAggregate (transform_aggregate_partial.rs:119-303): consumes blocks, emits one meta block
class TransformPartialAggregate : public AccumulatingTransform {
 public:
  explicit TransformPartialAggregate(AggregatorParams p) : table_(p) {}
  const char* name() const override { return "TransformPartialAggregate"; }
  std::vector<DataBlock> transform(DataBlock block) override { table_.add_groups(block); return {}; }
  std::vector<DataBlock> on_finish(bool output) override {
    if (!output) return {};
    DataBlock b; b.meta = std::make_shared<AggregateMetaSerialized>(table_.serialize());
    std::vector<DataBlock> v; v.push_back(std::move(b));
    return v;
  }
 private:
  AggregateHashTable table_;
};

// ---- the drop-in at the reference's block size (round 6) ------------------------------------------------------------------------
// The reference's pipeline hands every processor DataBlocks of <= max_block_size = 65,536 rows (settings_default.rs:142-148;
// TransformPartialAggregate::transform per block, transform_aggregate_partial.rs:262-270). A device call that size is a few
// microseconds of kernel: what a binding does about that decides whether the device's rate is reachable at all
// (profiles/r06_block_size_sweep.json). Two tools:
//   * TransformFusedPartialAggregate — TransformFilter -> CompoundBlockOperator::Map -> TransformPartialAggregate as ONE
//     operator over a PIPELINED table (dbhip_groupby_set_pipelined): a block costs one queued descriptor, 32 blocks one launch,
//     and the error a block may raise arrives at the checkpoint together with the number of blocks that were merged — the rest
//     is replayed here through the three separate operators, exactly what the synchronous call's caller does per block.
//   * BlockAccumulator — the squash in front of operators that have no pipelined form (the reference's own join build squashes
//     its input the same way, new_hash_join/memory/basic.rs:78-89): blocks are concatenated on the device up to `target_rows`.

// A filter predicate + one argument expression per aggregate, flattened post-order into ONE dbhip_expr_eval register program
// (16 registers, numeric / date / decimal / boolean nodes: what dbhip_groupby_add_block_program takes). Built once per pipeline.
class AggregateProgram {
 public:
  AggregateProgram(const std::optional<Expr>& filter, const std::vector<std::optional<Expr>>& args) {
    for (int r = 15; r >= 0; --r) free_.push_back(r);
    if (filter) { filter_reg_ = emit(*filter); keep(*filter, filter_reg_); }
    for (const auto& a : args) {
      if (!a) { arg_regs_.push_back(DBHIP_ARG_NONE); continue; }
      if (a->kind == Expr::ColumnRef) { arg_regs_.push_back(DBHIP_ARG_INPUT(input_of(a->id))); continue; }
      const int r = emit(*a);
      keep(*a, r);   // a result stays in its register; a later expression that contains it (Q1's charge contains disc_price) reads it there
      arg_regs_.push_back(r);
    }
  }
  bool ok() const { return ok_; }
  const std::vector<size_t>& inputs() const { return inputs_; }   // block column offsets, in program input order
  // the program over THIS block's columns
  void bind(const DataBlock& block, std::vector<dbhip_col>& cols, dbhip_agg_program& ap) const {
    cols.clear();
    for (size_t id : inputs_) cols.push_back(block.get_by_offset(id).c());
    ap.prog = ins_.empty() ? nullptr : ins_.data(); ap.n_ins = (int32_t)ins_.size();
    ap.inputs = cols.data(); ap.n_inputs = (int32_t)cols.size();
    ap.filter_reg = filter_reg_; ap.arg_regs = arg_regs_.data();
  }
 private:
  int input_of(size_t id) {
    for (size_t k = 0; k < inputs_.size(); ++k) if (inputs_[k] == id) return (int)k;
    if (inputs_.size() == 8) { ok_ = false; return 0; }
    inputs_.push_back(id);
    return (int)inputs_.size() - 1;
  }
  int alloc() { if (free_.empty()) { ok_ = false; return 0; } int r = free_.back(); free_.pop_back(); return r; }
  void release(int r) { if (!kept_regs_.count(r)) free_.push_back(r); }
  void keep(const Expr& e, int r) { kept_[e.sql_display() + ":" + e.type.name()] = r; kept_regs_.insert(r); }
  void ins(int op, int dst, int a, int b, const DataType& t, uint64_t imm) {
    dbhip_expr_ins i; memset(&i, 0, sizeof(i));
    i.op = op; i.dst = dst; i.a = a; i.b = b; i.type = t.id; i.imm = imm; i.precision = t.precision; i.scale = t.scale;
    ins_.push_back(i);
  }
  int emit(const Expr& e) {
    if (!ok_) return 0;
    const DataType& t = e.type;
    if (e.kind == Expr::FunctionCall) {
      auto k = kept_.find(e.sql_display() + ":" + t.name());
      if (k != kept_.end()) return k->second;
    }
    const bool plain = ((t.id >= DBHIP_T_BOOL && t.id <= DBHIP_T_TIMESTAMP) || t.is_decimal()) && t.dim == 0;
    if (!plain) { ok_ = false; return 0; }
    switch (e.kind) {
      case Expr::ColumnRef: { const int k = input_of(e.id); const int r = alloc(); ins(DBHIP_EX_LOAD, r, k, 0, t, 0); return r; }
      case Expr::Constant: {
        if (e.scalar.is_null) { ok_ = false; return 0; }
        uint64_t imm;
        if (t.id == DBHIP_T_F32) { double d = (double)(float)e.scalar.f; memcpy(&imm, &d, 8); }
        else if (t.id == DBHIP_T_F64) memcpy(&imm, &e.scalar.f, 8);
        else imm = (uint64_t)e.scalar.i;
        const int r = alloc(); ins(DBHIP_EX_CONST, r, 0, 0, t, imm); return r;
      }
      case Expr::Cast: {
        const int a = emit(e.args[0]);
        if (!ok_ || e.args[0].type.same_physical(t)) return a;
        ins(DBHIP_EX_CAST, a, a, 0, t, 0);
        return a;
      }
      default: break;
    }
    static const std::map<std::string, int> ops = {{"plus", DBHIP_EX_PLUS}, {"minus", DBHIP_EX_MINUS}, {"multiply", DBHIP_EX_MULTIPLY},
        {"divide", DBHIP_EX_DIVIDE}, {"eq", DBHIP_EX_EQ}, {"noteq", DBHIP_EX_NOTEQ}, {"lt", DBHIP_EX_LT}, {"lte", DBHIP_EX_LTE},
        {"gt", DBHIP_EX_GT}, {"gte", DBHIP_EX_GTE}};
    if (e.fname == "and_filters" || e.fname == "or_filters") {
      int acc = -1;
      for (const Expr& a : e.args) {
        const int r = emit(a);
        if (!ok_) return 0;
        ins(DBHIP_EX_IS_TRUE, r, r, 0, DataType::of(DBHIP_T_BOOL), 0);
        if (acc < 0) { acc = r; continue; }
        ins(e.fname == "or_filters" ? DBHIP_EX_OR : DBHIP_EX_AND, acc, acc, r, DataType::of(DBHIP_T_BOOL), 0);
        release(r);
      }
      return acc;
    }
    auto it = ops.find(e.fname);
    if (it == ops.end() || e.args.size() != 2) { ok_ = false; return 0; }
    const int a = emit(e.args[0]);
    const int b = emit(e.args[1]);
    if (!ok_) return 0;
    // a fresh destination: an operand may be a column another result needs as it is (the compiler eliminates the moves)
    const int d = alloc();
    ins(it->second, d, a, b, t, 0);
    release(a);
    if (b != a) release(b);
    return d;
  }
  std::vector<dbhip_expr_ins> ins_;
  std::vector<size_t> inputs_;
  std::vector<int32_t> arg_regs_;
  std::vector<int> free_;
  std::map<std::string, int> kept_;
  std::set<int> kept_regs_;
  int32_t filter_reg_ = -1;
  bool ok_ = true;
};

// BlockAccumulator: squashes incoming blocks on the device until `target_rows` rows are together (fixed-width and Boolean
// columns, validity, strings: dbhip_concat_columns). add() returns the squashed block when one is full; finish() the rest.
class BlockAccumulator {
 public:
  explicit BlockAccumulator(int64_t target_rows) : target_(target_rows) {}
  std::optional<DataBlock> add(DataBlock b) {
    if (b.num_rows == 0) return std::nullopt;
    rows_ += b.num_rows;
    parts_.push_back(std::move(b));
    if (rows_ < target_) return std::nullopt;
    return flush();
  }
  std::optional<DataBlock> finish() { return parts_.empty() ? std::nullopt : std::optional<DataBlock>(flush()); }
 private:
  DataBlock flush() {
    DataBlock out; out.num_rows = rows_;
    if (parts_.size() == 1) { out = std::move(parts_[0]); parts_.clear(); rows_ = 0; return out; }
    const size_t ncols = parts_[0].columns.size();
    for (size_t c = 0; c < ncols; ++c) {
      std::vector<dbhip_col> cols; std::vector<int64_t> rows;
      bool any_valid = false; int nbuf = 0;
      for (const DataBlock& p : parts_) { cols.push_back(p.columns[c].c()); rows.push_back(p.num_rows); any_valid |= (bool)p.columns[c].validity; nbuf += cols.back().n_buffers; }
      const Column& first = parts_[0].columns[c];
      Column r; r.type = first.type; r.len = rows_;
      const size_t words = (size_t)(rows_ + 63) / 64;
      r.data = make_buf(first.type.id == DBHIP_T_BOOL ? words * 8 + 8 : (size_t)rows_ * first.type.elem_size() + 16);
      if (any_valid) { r.validity = make_buf(words * 8 + 8); r.type.nullable = true; }
      if (first.type.id == DBHIP_T_STRING && nbuf > 0) throw ErrorCode::Unimplemented("BlockAccumulator: long strings (pass such blocks through)");
      int32_t nb_out = 0;
      check(dbhip_concat_columns(cols.data(), rows.data(), nullptr, (int32_t)cols.size(), r.data->ptr(), r.validity ? (uint8_t*)r.validity->ptr() : nullptr,
                                 nullptr, &nb_out, nullptr));
      out.columns.push_back(std::move(r));
    }
    check(dbhip_stream_sync(nullptr));   // the parts' buffers may go now
    parts_.clear(); rows_ = 0;
    return out;
  }
  int64_t target_, rows_ = 0;
  std::vector<DataBlock> parts_;
};

// TransformFilter -> CompoundBlockOperator::Map -> TransformPartialAggregate as one operator (see above). `filter`: the
// predicate over the INPUT block; `args[i]`: aggregate i's argument as an expression over the input block (nullopt = count(*)).
// Groups beyond the fused kernel's reach (more than 8 per workgroup), row errors and unsupported shapes fall back — block by
// block — to the three operators, which raise the reference's own error text where there is one.
class TransformFusedPartialAggregate : public AccumulatingTransform {
 public:
  TransformFusedPartialAggregate(AggregatorParams p, std::optional<Expr> filter, std::vector<std::optional<Expr>> args, bool pipelined = true,
                                 void* stream = nullptr)
      : params_(std::move(p)), filter_(std::move(filter)), args_(std::move(args)), prog_(filter_, args_), table_(params_), stream_(stream) {
    fused_ = prog_.ok();
    if (fused_ && pipelined) {
      const int32_t rc = dbhip_groupby_set_pipelined(table_.handle(), 1, stream_);
      if (rc == DBHIP_ERR_UNSUPPORTED) fused_ = false; else check(rc);
      pipelined_ = fused_;
    }
  }
  const char* name() const override { return "TransformFusedPartialAggregate"; }
  // PREPARE: the run-time specialised kernels of this pipeline's shape (dbhip_groupby_prepare_program), from one sample block
  void prepare(const DataBlock& sample) {
    if (!fused_) return;
    std::vector<dbhip_col> cols, keys; dbhip_agg_program ap;
    bind(sample, cols, keys, ap);
    const int32_t rc = dbhip_groupby_prepare_program(table_.handle(), keys.data(), &ap);
    if (rc == DBHIP_ERR_UNSUPPORTED || rc == DBHIP_ERR_INVALID) fused_ = false; else check(rc);
  }
  std::vector<DataBlock> transform(DataBlock block) override {
    if (block.num_rows == 0) return {};
    if (!fused_) { slow_path(block); return {}; }
    std::vector<dbhip_col> cols, keys; dbhip_agg_program ap;
    bind(block, cols, keys, ap);
    const int32_t rc = dbhip_groupby_add_block_program(table_.handle(), keys.data(), &ap, block.num_rows, nullptr, 0, stream_);
    if (rc == DBHIP_ERR_CAPACITY || rc == DBHIP_ERR_ROW_ERRORS || rc == DBHIP_ERR_UNSUPPORTED) {
      // (synchronous mode, or a shape the kernel refuses at the call) nothing of the block was merged
      if (rc != DBHIP_ERR_ROW_ERRORS) fused_ = false;
      drain();
      slow_path(block);
      return {};
    }
    check(rc);
    if (pipelined_) retained_.push_back(std::move(block));   // inputs of queued kernels: alive until the checkpoint
    return {};
  }
  std::vector<DataBlock> on_finish(bool output) override {
    drain();
    if (!output) return {};
    DataBlock b; b.meta = std::make_shared<AggregateMetaSerialized>(table_.serialize());
    std::vector<DataBlock> v; v.push_back(std::move(b));
    return v;
  }
  int64_t blocks_replayed() const { return replayed_; }
  AggregateHashTable& table() { return table_; }
 private:
  void bind(const DataBlock& block, std::vector<dbhip_col>& cols, std::vector<dbhip_col>& keys, dbhip_agg_program& ap) const {
    prog_.bind(block, cols, ap);
    keys.clear();
    for (size_t gcol : params_.group_columns) keys.push_back(block.get_by_offset(gcol).c());
  }
  // the checkpoint of a pipelined table: blocks [committed, queued) were not merged -> the three operators take them
  void drain() {
    if (!pipelined_ || retained_.empty()) return;
    int64_t committed = 0;
    const int32_t rc = dbhip_groupby_checkpoint(table_.handle(), &committed, stream_);
    std::vector<DataBlock> blocks;
    blocks.swap(retained_);
    if (rc == DBHIP_OK) return;
    if (rc != DBHIP_ERR_CAPACITY && rc != DBHIP_ERR_ROW_ERRORS && rc != DBHIP_ERR_UNSUPPORTED) check(rc);
    if (rc != DBHIP_ERR_ROW_ERRORS) fused_ = false;   // the keys (or the shape) are not for this kernel: the rest of the stream goes the slow way
    for (size_t i = (size_t)committed; i < blocks.size(); ++i) { slow_path(blocks[i]); ++replayed_; }
  }
  // TransformFilter -> maps -> add_groups, one kernel per node: raises the reference's row errors where the fused kernel only counted them
  void slow_path(const DataBlock& block) {
    DataBlock b = filter_ ? FilterExecutor(*filter_).filter(block) : block;
    if (b.num_rows == 0) return;
    std::vector<dbhip_col> keys, aggs(args_.size());
    std::vector<Column> hold;
    hold.reserve(args_.size());
    for (size_t gcol : params_.group_columns) keys.push_back(b.get_by_offset(gcol).c());
    for (size_t a = 0; a < args_.size(); ++a) {
      memset(&aggs[a], 0, sizeof(dbhip_col));
      if (!args_[a]) continue;
      Evaluator ev(b);
      Value v = ev.run(*args_[a]);
      if (v.is_scalar) throw ErrorCode::Unimplemented("constant aggregate argument");
      hold.push_back(v.column);
      aggs[a] = hold.back().c();
    }
    check(dbhip_groupby_add_block(table_.handle(), keys.data(), aggs.data(), b.num_rows, stream_));
  }
  AggregatorParams params_;
  std::optional<Expr> filter_;
  std::vector<std::optional<Expr>> args_;
  AggregateProgram prog_;
  AggregateHashTable table_;
  void* stream_;
  bool fused_ = false, pipelined_ = false;
  std::vector<DataBlock> retained_;
  int64_t replayed_ = 0;
};

// TransformFinalAggregate (transform_aggregate_final.rs:69-551): merges partial states, emits the result
class TransformFinalAggregate : public AccumulatingTransform {
 public:
  explicit TransformFinalAggregate(AggregatorParams p) : table_(p) {}
  const char* name() const override { return "TransformFinalAggregate"; }
  std::vector<DataBlock> transform(DataBlock block) override {
    auto m = std::static_pointer_cast<AggregateMetaSerialized>(block.meta);
    if (!m) throw ErrorCode::Internal("TransformFinalAggregate expects AggregateMeta::Serialized");
    table_.combine(*m);
    return {};
  }
  std::vector<DataBlock> on_finish(bool output) override {
    if (!output) return {};
    std::vector<DataBlock> v; v.push_back(table_.merge_result());
    return v;
  }
 private:
  AggregateHashTable table_;
};

// PartialSingleStateAggregator + FinalSingleStateAggregator (transform_single_key.rs:43-279): no GROUP BY,
// `accumulate` of every block into one state per function (sum -> dbhip_sum, count -> validity popcount)
class SingleStateAggregator : public AccumulatingTransform {
 public:
  explicit SingleStateAggregator(std::vector<AggregateFunctionDesc> f) : funcs_(std::move(f)), i_(funcs_.size(), 0), f_(funcs_.size(), 0.0) {}
  const char* name() const override { return "AggregatorPartialTransform"; }
  std::vector<DataBlock> transform(DataBlock block) override {
    for (size_t a = 0; a < funcs_.size(); ++a) {
      const auto& fn = funcs_[a];
      if (fn.name == "count") {
        if (!fn.arg || !block.get_by_offset(*fn.arg).validity) i_[a] += (uint64_t)block.num_rows;
        else i_[a] += (uint64_t)count_bits(block.get_by_offset(*fn.arg).validity, block.num_rows);
      } else if (fn.name == "sum") {
        dbhip_col c = block.get_by_offset(*fn.arg).c();
        Buf out = make_buf(8); out->fill(0);
        check(dbhip_sum(&c, block.num_rows, out->ptr(), nullptr));
        if (fn.arg_type.id == DBHIP_T_F32 || fn.arg_type.id == DBHIP_T_F64) { double d; out->download(&d, 8); f_[a] += d; }
        else { uint64_t u; out->download(&u, 8); i_[a] += u; }  // wrapping, like NumberSumState (aggregate_sum.rs:113-129)
      } else throw ErrorCode::Unimplemented("single-state `" + fn.name + "`");
    }
    return {};
  }
  std::vector<DataBlock> on_finish(bool) override { return {}; }
  int64_t int_result(size_t a) const { return (int64_t)i_[a]; }
  double float_result(size_t a) const { return f_[a]; }
 private:
  std::vector<AggregateFunctionDesc> funcs_;
  std::vector<uint64_t> i_;
  std::vector<double> f_;
};

// ---- hash join (new_hash_join/join.rs:22-53) -----------------------------------------------------
struct JoinStream {
  virtual ~JoinStream() = default;
  virtual std::optional<DataBlock> next() = 0;
};
struct Join {
  virtual ~Join() = default;
  virtual void add_block(std::optional<DataBlock> data) = 0;
  virtual void final_build() = 0;
  virtual std::unique_ptr<JoinStream> probe_block(DataBlock data) = 0;
};

// Another conjunct on top of the key equality (the `CONJUNCT = true` streams: inner_join.rs:278-310 InnerHashJoinFilterStream,
// left_join.rs:262-292, left_join_semi.rs:320-350, left_join_anti.rs:270-300, right_join.rs:256-290): a predicate over the JOINED
// rows (probe columns ++ build columns) -> Boolean column; NULL drops the pair like FALSE. A probe / build row all of whose
// pairs were dropped counts as unmatched.
using JoinConjunct = std::function<Column(const DataBlock&)>;
struct JoinPairs { Buf pi, bi; int64_t count = 0; };
inline JoinPairs filter_pairs(const DataBlock& probe, const DataBlock& build, JoinPairs p, const JoinConjunct& conj) {
  if (!conj || p.count == 0) return p;
  DataBlock j = take_block(probe, p.pi, p.count);
  DataBlock b = take_block(build, p.bi, p.count);
  for (auto& c : b.columns) j.columns.push_back(c);
  j.num_rows = p.count;
  Selection s = filter_select(conj(j));
  JoinPairs o; o.count = s.count;
  o.pi = make_buf((size_t)s.count * 4 + 64); o.bi = make_buf((size_t)s.count * 4 + 64);
  check(dbhip_take(p.pi->ptr(), 4, (const uint32_t*)s.sel->ptr(), s.count, o.pi->ptr(), nullptr));
  check(dbhip_take(p.bi->ptr(), 4, (const uint32_t*)s.sel->ptr(), s.count, o.bi->ptr(), nullptr));
  return o;
}
// Boolean column: probe row kept at least one pair (dbhip_bitmap_set_indices over the surviving pairs' probe rows)
inline Column rows_with_pairs(const JoinPairs& p, int64_t n) {
  Column m; m.type = DataType::of(DBHIP_T_BOOL); m.len = n;
  m.data = make_buf((size_t)(n + 63) / 64 * 8 + 64); m.data->fill(0);
  check(dbhip_bitmap_set_indices((const uint32_t*)p.pi->ptr(), p.count, (uint8_t*)m.data->ptr(), n, nullptr));
  return m;
}
inline Column not_bitmap(const Column& m) {   // Boolean equality with a false constant, like the reference's `not`
  Column u = m;
  u.data = make_buf((size_t)(m.len + 63) / 64 * 8 + 64);
  Column f = Column::from_bools(std::vector<bool>{false});
  dbhip_col a = m.c(), b = f.c();
  b.is_scalar = 1;
  check(dbhip_cmp(DBHIP_CMP_EQ, &a, &b, m.len, (uint8_t*)u.data->ptr(), nullptr));
  return u;
}

// InnerHashJoin on one u64/i64 key (memory/inner_join.rs:47-271): output = probe columns ++ build columns
class InnerHashJoin : public Join {
 public:
  InnerHashJoin(size_t build_key, size_t probe_key, int64_t max_block_size = 65536) : bk_(build_key), pk_(probe_key), max_block_(max_block_size) {
    check(dbhip_join_create(1024, &h_));
  }
  ~InnerHashJoin() override { if (h_) dbhip_join_destroy(h_); }
  void add_block(std::optional<DataBlock> data) override {
    if (!data) return;
    if (!chunks_.empty()) throw ErrorCode::Unimplemented("InnerHashJoin host mirror keeps one build chunk (concat the build side first)");
    const Column& k = data->get_by_offset(bk_);
    check(dbhip_join_add_build(h_, (const uint64_t*)k.data->ptr(), k.validity ? (const uint8_t*)k.validity->ptr() : nullptr, k.len, nullptr));
    chunks_.push_back(std::move(*data));
  }
  void final_build() override { check(dbhip_join_finalize(h_, nullptr)); }
  std::unique_ptr<JoinStream> probe_block(DataBlock data) override {
    const Column& k = data.get_by_offset(pk_);
    const uint8_t* v = k.validity ? (const uint8_t*)k.validity->ptr() : nullptr;
    uint64_t total = 0;
    check(dbhip_join_probe_count(h_, (const uint64_t*)k.data->ptr(), v, k.len, &total, nullptr));
    Buf pi = make_buf((size_t)total * 4), bi = make_buf((size_t)total * 4);
    uint64_t got = 0;
    check(dbhip_join_probe(h_, (const uint64_t*)k.data->ptr(), v, k.len, (uint32_t*)pi->ptr(), (uint32_t*)bi->ptr(), (int64_t)total, &got, nullptr));
    JoinPairs p{pi, bi, (int64_t)got};
    if (conj_ && !chunks_.empty()) p = filter_pairs(data, chunks_[0], p, conj_);
    return std::make_unique<Stream>(std::move(data), chunks_.empty() ? DataBlock() : chunks_[0], p.pi, p.bi, p.count, max_block_);
  }
  void set_conjunct(JoinConjunct c) { conj_ = std::move(c); }
 private:
  JoinConjunct conj_;
  struct Stream : JoinStream {
    DataBlock probe, build; Buf pi, bi; int64_t total, pos = 0, max_block;
    Stream(DataBlock p, DataBlock b, Buf pi_, Buf bi_, int64_t t, int64_t mb) : probe(std::move(p)), build(std::move(b)), pi(pi_), bi(bi_), total(t), max_block(mb) {}
    std::optional<DataBlock> next() override {
      if (pos >= total) return std::nullopt;
      int64_t k = total - pos < max_block ? total - pos : max_block;
      DataBlock out = take_block(probe, (const uint32_t*)pi->ptr() + pos, k);
      DataBlock b = take_block(build, (const uint32_t*)bi->ptr() + pos, k);
      for (auto& c : b.columns) out.columns.push_back(c);
      pos += k;
      return out;
    }
  };
  size_t bk_, pk_;
  int64_t max_block_;
  dbhip_join* h_ = nullptr;
  std::vector<DataBlock> chunks_;
};

// Left-outer / left-semi / left-anti joins on one u64/i64 key, no other conjunct (memory/left_join.rs:185-260,
// left_join_semi.rs, left_join_anti.rs): the matched Bitmap (dbhip_join_probe_mark) selects the semi / anti rows; the outer
// join emits the matched pairs with the build columns wrapped in a true validity and then ONE block of the unmatched probe rows
// with a null build block (dbhip_take_outer with row id 0xFFFFFFFF).
enum class LeftJoinKind { Outer, Semi, Anti };
class LeftHashJoin : public Join {
 public:
  LeftHashJoin(LeftJoinKind kind, size_t build_key, size_t probe_key) : kind_(kind), bk_(build_key), pk_(probe_key) { check(dbhip_join_create(1024, &h_)); }
  ~LeftHashJoin() override { if (h_) dbhip_join_destroy(h_); }
  void add_block(std::optional<DataBlock> data) override {
    if (!data) return;
    if (!chunks_.empty()) throw ErrorCode::Unimplemented("LeftHashJoin host mirror keeps one build chunk (concat the build side first)");
    const Column& k = data->get_by_offset(bk_);
    check(dbhip_join_add_build(h_, (const uint64_t*)k.data->ptr(), k.validity ? (const uint8_t*)k.validity->ptr() : nullptr, k.len, nullptr));
    chunks_.push_back(std::move(*data));
  }
  void final_build() override { check(dbhip_join_finalize(h_, nullptr)); }
  std::unique_ptr<JoinStream> probe_block(DataBlock data) override {
    const Column& k = data.get_by_offset(pk_);
    const uint8_t* v = k.validity ? (const uint8_t*)k.validity->ptr() : nullptr;
    const int64_t n = k.len;
    if (conj_ && !chunks_.empty()) return probe_block_conjunct(std::move(data), k, v, n);
    // matched Bitmap -> the rows of the semi / anti result, or the unmatched tail of the outer result
    Column matched; matched.type = DataType::of(DBHIP_T_BOOL); matched.len = n;
    matched.data = make_buf((size_t)(n + 63) / 64 * 8 + 64); matched.data->fill(0);
    uint64_t nm = 0;
    check(dbhip_join_probe_mark(h_, (const uint64_t*)k.data->ptr(), v, n, (uint8_t*)matched.data->ptr(), &nm, nullptr));
    if (kind_ == LeftJoinKind::Semi) {
      Selection s = filter_select(matched);
      return std::make_unique<Once>(take_block(data, s.sel, s.count));
    }
    Column unmatched = matched;
    unmatched.data = make_buf((size_t)(n + 63) / 64 * 8 + 64);
    {  // NOT matched: Boolean equality with a false constant, like the reference's `not`
      Column f = Column::from_bools(std::vector<bool>{false});
      dbhip_col a = matched.c(), b = f.c();
      b.is_scalar = 1;
      check(dbhip_cmp(DBHIP_CMP_EQ, &a, &b, n, (uint8_t*)unmatched.data->ptr(), nullptr));
    }
    Selection us = filter_select(unmatched);
    if (kind_ == LeftJoinKind::Anti) return std::make_unique<Once>(take_block(data, us.sel, us.count));
    uint64_t total = 0, got = 0;
    check(dbhip_join_probe_count(h_, (const uint64_t*)k.data->ptr(), v, n, &total, nullptr));
    const int64_t rows = (int64_t)total + us.count;
    Buf pi = make_buf((size_t)rows * 4 + 64), bi = make_buf((size_t)rows * 4 + 64);
    check(dbhip_join_probe(h_, (const uint64_t*)k.data->ptr(), v, n, (uint32_t*)pi->ptr(), (uint32_t*)bi->ptr(), (int64_t)total, &got, nullptr));
    check(dbhip_memcpy_d2d((uint32_t*)pi->ptr() + total, us.sel->ptr(), (size_t)us.count * 4, nullptr));
    check(dbhip_memset((uint32_t*)bi->ptr() + total, 0xFF, (size_t)us.count * 4, nullptr));
    DataBlock out = take_block(data, pi, rows);
    if (!chunks_.empty())
      for (const Column& c : chunks_[0].columns) {
        Column r; r.type = c.type; r.type.nullable = true; r.len = rows;
        const size_t es = c.type.elem_size();
        if (c.type.id == DBHIP_T_BOOL || (es != 1 && es != 2 && es != 4 && es != 8 && es != 16)) throw ErrorCode::Unimplemented("outer join build column " + c.type.name());
        r.data = make_buf((size_t)rows * es + 64);
        r.validity = make_buf((size_t)(rows + 63) / 64 * 8 + 8);
        check(dbhip_take_outer(c.data->ptr(), c.validity ? (const uint8_t*)c.validity->ptr() : nullptr, 0, (int32_t)es, (const uint32_t*)bi->ptr(), rows,
                               r.data->ptr(), (uint8_t*)r.validity->ptr(), nullptr));
        out.columns.push_back(r);
      }
    return std::make_unique<Once>(std::move(out));
  }
  void set_conjunct(JoinConjunct c) { conj_ = std::move(c); }
 private:
  // CONJUNCT = true (left_join.rs:262-292, left_join_semi.rs, left_join_anti.rs): the key matches are filtered first, the rows that
  // kept a pair are the matched ones
  std::unique_ptr<JoinStream> probe_block_conjunct(DataBlock data, const Column& k, const uint8_t* v, int64_t n) {
    uint64_t total = 0, got = 0;
    check(dbhip_join_probe_count(h_, (const uint64_t*)k.data->ptr(), v, n, &total, nullptr));
    JoinPairs p; p.pi = make_buf((size_t)total * 4 + 64); p.bi = make_buf((size_t)total * 4 + 64);
    check(dbhip_join_probe(h_, (const uint64_t*)k.data->ptr(), v, n, (uint32_t*)p.pi->ptr(), (uint32_t*)p.bi->ptr(), (int64_t)total, &got, nullptr));
    p.count = (int64_t)got;
    p = filter_pairs(data, chunks_[0], p, conj_);
    Column matched = rows_with_pairs(p, n);
    if (kind_ == LeftJoinKind::Semi) {
      Selection s = filter_select(matched);
      return std::make_unique<Once>(take_block(data, s.sel, s.count));
    }
    Selection us = filter_select(not_bitmap(matched));
    if (kind_ == LeftJoinKind::Anti) return std::make_unique<Once>(take_block(data, us.sel, us.count));
    const int64_t rows = p.count + us.count;
    Buf pi = make_buf((size_t)rows * 4 + 64), bi = make_buf((size_t)rows * 4 + 64);
    check(dbhip_memcpy_d2d(pi->ptr(), p.pi->ptr(), (size_t)p.count * 4, nullptr));
    check(dbhip_memcpy_d2d(bi->ptr(), p.bi->ptr(), (size_t)p.count * 4, nullptr));
    check(dbhip_memcpy_d2d((uint32_t*)pi->ptr() + p.count, us.sel->ptr(), (size_t)us.count * 4, nullptr));
    check(dbhip_memset((uint32_t*)bi->ptr() + p.count, 0xFF, (size_t)us.count * 4, nullptr));
    DataBlock out = take_block(data, pi, rows);
    for (const Column& c : chunks_[0].columns) {
      Column r; r.type = c.type; r.type.nullable = true; r.len = rows;
      const size_t es = c.type.elem_size();
      if (c.type.id == DBHIP_T_BOOL || (es != 1 && es != 2 && es != 4 && es != 8 && es != 16)) throw ErrorCode::Unimplemented("outer join build column " + c.type.name());
      r.data = make_buf((size_t)rows * es + 64);
      r.validity = make_buf((size_t)(rows + 63) / 64 * 8 + 8);
      check(dbhip_take_outer(c.data->ptr(), c.validity ? (const uint8_t*)c.validity->ptr() : nullptr, 0, (int32_t)es, (const uint32_t*)bi->ptr(), rows,
                             r.data->ptr(), (uint8_t*)r.validity->ptr(), nullptr));
      out.columns.push_back(r);
    }
    out.num_rows = rows;
    return std::make_unique<Once>(std::move(out));
  }
  JoinConjunct conj_;
  struct Once : JoinStream {
    std::optional<DataBlock> b;
    explicit Once(DataBlock x) : b(std::move(x)) {}
    std::optional<DataBlock> next() override { auto r = std::move(b); b.reset(); return r; }
  };
  LeftJoinKind kind_;
  size_t bk_, pk_;
  dbhip_join* h_ = nullptr;
  std::vector<DataBlock> chunks_;
};

// Right-outer / right-semi / right-anti / full-outer joins on one u64/i64 key, no other conjunct (memory/right_join.rs,
// right_join_semi.rs, right_join_anti.rs, full_join.rs): every probe block emits its matched pairs (right / full; full also the
// unmatched probe rows with a null build block, like LeftHashJoin) and ORs the build rows it reached into the table's scan map
// (dbhip_join_mark_build); after the LAST probe block final_probe() emits the build rows the map says were matched (semi), were
// not (anti), or were not with a NULL probe side (right / full) — the reference's `final_probe` / scan-map pass.
enum class RightJoinKind { Outer, Semi, Anti, Full };
class RightHashJoin : public Join {
 public:
  RightHashJoin(RightJoinKind kind, size_t build_key, size_t probe_key) : kind_(kind), bk_(build_key), pk_(probe_key) { check(dbhip_join_create(1024, &h_)); }
  ~RightHashJoin() override { if (h_) dbhip_join_destroy(h_); }
  void add_block(std::optional<DataBlock> data) override {
    if (!data) return;
    if (!chunks_.empty()) throw ErrorCode::Unimplemented("RightHashJoin host mirror keeps one build chunk (concat the build side first)");
    const Column& k = data->get_by_offset(bk_);
    check(dbhip_join_add_build(h_, (const uint64_t*)k.data->ptr(), k.validity ? (const uint8_t*)k.validity->ptr() : nullptr, k.len, nullptr));
    chunks_.push_back(std::move(*data));
  }
  void final_build() override { check(dbhip_join_finalize(h_, nullptr)); }
  std::unique_ptr<JoinStream> probe_block(DataBlock data) override {
    const Column& k = data.get_by_offset(pk_);
    const uint8_t* v = k.validity ? (const uint8_t*)k.validity->ptr() : nullptr;
    const int64_t n = k.len;
    if (probe_types_.empty()) for (const Column& c : data.columns) probe_types_.push_back(c.type);
    uint64_t total = 0, got = 0;
    check(dbhip_join_probe_count(h_, (const uint64_t*)k.data->ptr(), v, n, &total, nullptr));
    Buf pi = make_buf((size_t)total * 4 + 64), bi = make_buf((size_t)total * 4 + 64);
    check(dbhip_join_probe(h_, (const uint64_t*)k.data->ptr(), v, n, (uint32_t*)pi->ptr(), (uint32_t*)bi->ptr(), (int64_t)total, &got, nullptr));
    if (conj_ && !chunks_.empty()) {   // CONJUNCT = true (right_join.rs:256-290): only the pairs that pass reach the scan map
      JoinPairs p = filter_pairs(data, chunks_[0], JoinPairs{pi, bi, (int64_t)got}, conj_);
      pi = p.pi; bi = p.bi; got = (uint64_t)p.count;
    }
    check(dbhip_join_mark_build(h_, (const uint32_t*)bi->ptr(), (int64_t)got, nullptr));
    if (kind_ == RightJoinKind::Semi || kind_ == RightJoinKind::Anti) return std::make_unique<Once>(std::nullopt);   // everything comes from final_probe
    // matched pairs: probe columns become Nullable with a true validity (the probe side is the nullable one of a right join)
    DataBlock out = take_block(data, pi, (int64_t)got);
    for (Column& c : out.columns) wrap_true_validity(c);
    DataBlock b = chunks_.empty() ? DataBlock() : take_block(chunks_[0], bi, (int64_t)got);
    if (kind_ == RightJoinKind::Full) for (Column& c : b.columns) wrap_true_validity(c);
    for (auto& c : b.columns) out.columns.push_back(c);
    out.num_rows = (int64_t)got;
    if (kind_ != RightJoinKind::Full) return std::make_unique<Once>(std::move(out));
    // full: + the unmatched probe rows with a null build block, as a second block
    Column matched; matched.type = DataType::of(DBHIP_T_BOOL); matched.len = n;
    matched.data = make_buf((size_t)(n + 63) / 64 * 8 + 64); matched.data->fill(0);
    uint64_t nm = 0;
    if (conj_ && !chunks_.empty()) check(dbhip_bitmap_set_indices((const uint32_t*)pi->ptr(), (int64_t)got, (uint8_t*)matched.data->ptr(), n, nullptr));
    else check(dbhip_join_probe_mark(h_, (const uint64_t*)k.data->ptr(), v, n, (uint8_t*)matched.data->ptr(), &nm, nullptr));
    Selection us = filter_select(not_bitmap(matched));
    DataBlock tail = take_block(data, us.sel, us.count);
    for (Column& c : tail.columns) wrap_true_validity(c);
    if (!chunks_.empty()) for (const Column& c : chunks_[0].columns) tail.columns.push_back(null_column(c.type, us.count));
    tail.num_rows = us.count;
    return std::make_unique<Two>(std::move(out), std::move(tail));
  }
  void set_conjunct(JoinConjunct c) { conj_ = std::move(c); }
  // after the last probe block (Join::final_probe): build rows by the scan map
  std::optional<DataBlock> final_probe() {
    if (chunks_.empty()) return std::nullopt;
    int64_t rows = 0;
    check(dbhip_join_build_matched(h_, nullptr, &rows, nullptr));
    Column m; m.type = DataType::of(DBHIP_T_BOOL); m.len = rows;
    m.data = make_buf((size_t)(rows + 63) / 64 * 8 + 64); m.data->fill(0);
    check(dbhip_join_build_matched(h_, (uint8_t*)m.data->ptr(), &rows, nullptr));
    Column want = m;
    if (kind_ != RightJoinKind::Semi) {   // the UNMATCHED build rows
      want.data = make_buf((size_t)(rows + 63) / 64 * 8 + 64);
      Column f = Column::from_bools(std::vector<bool>{false});
      dbhip_col a = m.c(), b = f.c();
      b.is_scalar = 1;
      check(dbhip_cmp(DBHIP_CMP_EQ, &a, &b, rows, (uint8_t*)want.data->ptr(), nullptr));
    }
    Selection s = filter_select(want);
    DataBlock b = take_block(chunks_[0], s.sel, s.count);
    if (kind_ == RightJoinKind::Semi || kind_ == RightJoinKind::Anti) return b;
    DataBlock out;   // right / full: a NULL probe side in front of the unmatched build rows
    for (const DataType& t : probe_types_) out.columns.push_back(null_column(t, s.count));
    if (kind_ == RightJoinKind::Full) for (Column& c : b.columns) wrap_true_validity(c);
    for (auto& c : b.columns) out.columns.push_back(c);
    out.num_rows = s.count;
    return out;
  }
 private:
  static void wrap_true_validity(Column& c) {   // wrap_true_validity (left_join.rs:243-249)
    if (!c.validity) { c.validity = const_bitmap(true, c.len); }
    c.type.nullable = true;
  }
  static Column null_column(DataType t, int64_t n) {
    Column c; c.type = t; c.type.nullable = true; c.len = n;
    const size_t es = t.id == DBHIP_T_BOOL ? 1 : t.elem_size();
    c.data = make_buf((size_t)(n > 0 ? n : 1) * (es ? es : 16) + 64); c.data->fill(0);
    c.validity = const_bitmap(false, n > 0 ? n : 1);
    return c;
  }
  struct Once : JoinStream {
    std::optional<DataBlock> b;
    explicit Once(std::optional<DataBlock> x) : b(std::move(x)) {}
    std::optional<DataBlock> next() override { auto r = std::move(b); b.reset(); return r; }
  };
  struct Two : JoinStream {
    std::optional<DataBlock> a, b;
    Two(DataBlock x, DataBlock y) : a(std::move(x)), b(std::move(y)) {}
    std::optional<DataBlock> next() override {
      if (a) { auto r = std::move(a); a.reset(); return r; }
      auto r = std::move(b); b.reset(); return r;
    }
  };
  JoinConjunct conj_;
  RightJoinKind kind_;
  size_t bk_, pk_;
  dbhip_join* h_ = nullptr;
  std::vector<DataBlock> chunks_;
  std::vector<DataType> probe_types_;
};

// ---- vector-cluster KMeans (kmeans.rs:93-291, TransformVectorCluster) ------------------------------------------------------------
struct KMeansResult { std::vector<uint32_t> assignments; std::vector<float> distances; int64_t k = 0; int32_t iterations = 0; };
// distance_type: 0 = L1, 1 = L2, 2 = Dot (VectorDistanceType); `column` = Vector(dim) of Float32
inline KMeansResult kmeans(int32_t distance_type, const Column& column, int64_t rows_per_cluster, bool normalize_input) {
  KMeansResult r;
  const int64_t n = column.len;
  Buf a = make_buf((size_t)(n > 0 ? n : 1) * 4 + 64), d = make_buf((size_t)(n > 0 ? n : 1) * 4 + 64);
  check(dbhip_kmeans(distance_type, (const float*)column.data->ptr(), n, (int32_t)column.type.dim, rows_per_cluster, normalize_input ? 1 : 0,
                     (uint32_t*)a->ptr(), (float*)d->ptr(), &r.k, &r.iterations, nullptr));
  r.assignments.resize((size_t)n); r.distances.resize((size_t)n);
  if (n) { a->download(r.assignments.data(), (size_t)n * 4); d->download(r.distances.data(), (size_t)n * 4); }
  return r;
}

// ---- HNSW vector index (hnsw_index/hnsw.rs:62-315) ------------------------------------------------------------------
class HNSWIndex {
 public:
  // HNSWIndex::build (hnsw.rs:142-305): `column` = Vector(dim) of Float32; distance: DBHIP_VEC_COSINE / L1 / L2
  static HNSWIndex build(size_t m, size_t ef_construct, const Column& column, int32_t distance, uint64_t seed = 1) {
    HNSWIndex ix;
    ix.dim_ = column.type.dim; ix.n_ = column.len; ix.m_ = m;
    check(dbhip_hnsw_build((const float*)column.data->ptr(), column.len, (int32_t)ix.dim_, distance, (int32_t)m, (int32_t)ef_construct, seed, &ix.h_, nullptr));
    return ix;
  }
  // the deterministic build (given levels, sequential insertion, the reference's summation order): dbhip_hnsw_build_sequential
  static HNSWIndex build_sequential(size_t m, size_t ef_construct, const Column& column, int32_t distance, const std::vector<int32_t>& levels) {
    HNSWIndex ix;
    ix.dim_ = column.type.dim; ix.n_ = column.len; ix.m_ = m;
    check(dbhip_hnsw_build_sequential((const float*)column.data->ptr(), column.len, (int32_t)ix.dim_, distance, (int32_t)m, (int32_t)ef_construct,
                                      levels.data(), &ix.h_, nullptr));
    return ix;
  }
  // what HNSWIndex::build hands to the index writer and HNSWIndex::open gets back (hnsw.rs:62-98, 237-300): the graph, the quantiser's
  // metadata and the encoded vectors; the byte formats of the four Binary columns are databend_amd/hnsw_format.py's
  struct Stored {
    std::vector<int32_t> levels; std::vector<uint32_t> links; std::vector<int32_t> nlinks; uint32_t entry_point = 0; int32_t entry_level = 0;
    float alpha = 0, offset = 0, multiplier = 0; int32_t actual_dim = 0; std::vector<uint8_t> encoded;
  };
  Stored store() const {
    Stored s;
    s.levels.resize((size_t)(n_ > 0 ? n_ : 1));
    int64_t nlists = 0;
    check(dbhip_hnsw_export_graph(h_, s.levels.data(), nullptr, nullptr, &nlists, &s.entry_point, &s.entry_level, nullptr));
    s.levels.resize((size_t)n_);
    s.nlinks.resize((size_t)(nlists > 0 ? nlists : 1));
    check(dbhip_hnsw_export_graph(h_, nullptr, nullptr, s.nlinks.data(), nullptr, nullptr, nullptr, nullptr));
    s.nlinks.resize((size_t)nlists);
    int64_t total = 0;
    for (int32_t c : s.nlinks) total += c;
    s.links.resize((size_t)(total > 0 ? total : 1));
    check(dbhip_hnsw_export_graph(h_, nullptr, s.links.data(), nullptr, nullptr, nullptr, nullptr, nullptr));
    s.links.resize((size_t)total);
    check(dbhip_hnsw_meta(h_, &s.alpha, &s.offset, &s.multiplier, &s.actual_dim));
    Buf enc = make_buf((size_t)(n_ > 0 ? n_ : 1) * (size_t)(s.actual_dim + 4) + 64);
    check(dbhip_hnsw_encoded(h_, enc->ptr(), nullptr));
    s.encoded.resize((size_t)n_ * (size_t)(s.actual_dim + 4));
    if (!s.encoded.empty()) enc->download(s.encoded.data(), s.encoded.size());
    return s;
  }
  // HNSWIndex::open (hnsw.rs:62-98): no original vectors, the index only searches
  static HNSWIndex open(int32_t distance, size_t dim, size_t m, const Stored& s) {
    HNSWIndex ix;
    ix.dim_ = dim; ix.n_ = (int64_t)s.levels.size(); ix.m_ = m;
    Buf enc = make_buf(s.encoded.size() + 64);
    if (!s.encoded.empty()) enc->upload(s.encoded.data(), s.encoded.size());
    const uint32_t zero = 0; const int32_t zero_i = 0;
    check(dbhip_hnsw_open((const uint8_t*)enc->ptr(), s.alpha, s.offset, s.multiplier, ix.n_, (int32_t)dim, distance, (int32_t)m, s.levels.empty() ? &zero_i : s.levels.data(),
                          s.links.empty() ? &zero : s.links.data(), s.nlinks.empty() ? &zero_i : s.nlinks.data(), s.entry_point, s.entry_level, &ix.h_, nullptr));
    check(dbhip_stream_sync(nullptr));   // `enc` is only read while the index is being made
    return ix;
  }
  HNSWIndex(HNSWIndex&& o) noexcept : h_(o.h_), dim_(o.dim_), n_(o.n_), m_(o.m_) { o.h_ = nullptr; }
  HNSWIndex(const HNSWIndex&) = delete;
  ~HNSWIndex() { if (h_) dbhip_hnsw_destroy(h_); }
  // HNSWIndex::search (hnsw.rs:100-118), ef = 4 * limit inside: -> (row ids u32 [nq][limit], distances f32 [nq][limit])
  std::pair<std::vector<uint32_t>, std::vector<float>> search(size_t limit, const Column& queries) const {
    Buf ids = make_buf((size_t)queries.len * limit * 4 + 16), dist = make_buf((size_t)queries.len * limit * 4 + 16);
    check(dbhip_hnsw_search(h_, (const float*)queries.data->ptr(), (int32_t)queries.len, (int32_t)limit, (uint32_t*)ids->ptr(), (float*)dist->ptr(), nullptr));
    std::vector<uint32_t> i((size_t)queries.len * limit); std::vector<float> d(i.size());
    ids->download(i.data(), i.size() * 4); dist->download(d.data(), d.size() * 4);
    return {i, d};
  }
 private:
  HNSWIndex() = default;
  dbhip_hnsw* h_ = nullptr;
  size_t dim_ = 0;
  int64_t n_ = 0;
  size_t m_ = 0;
};

// ---- sort (kernels/sort.rs:91-113) -----------------------------------------------------------------
struct SortColumnDescription { size_t offset; bool asc = true; bool nulls_first = false; };

inline DataBlock sort_block(const DataBlock& block, const std::vector<SortColumnDescription>& desc, std::optional<int64_t> limit = std::nullopt) {
  std::vector<dbhip_col> keys; std::vector<uint8_t> d, nf;
  for (auto& s : desc) { keys.push_back(block.get_by_offset(s.offset).c()); d.push_back(s.asc ? 0 : 1); nf.push_back(s.nulls_first ? 1 : 0); }
  int64_t m = limit && *limit < block.num_rows ? *limit : block.num_rows;
  Buf perm = make_buf((size_t)(m > 0 ? m : 1) * 4);
  check(dbhip_sort_perm(keys.data(), d.data(), nf.data(), (int32_t)keys.size(), block.num_rows, limit ? *limit : 0, (uint32_t*)perm->ptr(), nullptr));
  return take_block(block, perm, m);
}

// ---- exchange scatters (servers/flight/v1/scatter/flight_scatter.rs: trait FlightScatter { fn execute(&self, DataBlock) -> Vec<DataBlock> }) ----
struct FlightScatter {
  virtual ~FlightScatter() = default;
  virtual const char* name() const = 0;
  virtual std::vector<DataBlock> execute(const DataBlock& data_block) = 0;
};
// DataBlock::scatter(block, indices, scatter_size): rows grouped by index (stable) — one radix pass over the indices + one
// dbhip_take_block per 8 columns; `counts` = rows per destination (host)
inline std::vector<DataBlock> scatter_block(const DataBlock& block, const Buf& indices, const std::vector<uint64_t>& counts) {
  const int64_t n = block.num_rows;
  std::vector<DataBlock> out;
  Column idx; idx.type = DataType::of(DBHIP_T_U32); idx.len = n; idx.data = indices;
  Buf perm = make_buf((size_t)(n > 0 ? n : 1) * 4 + 64);
  if (n) {
    dbhip_col key = idx.c();
    const uint8_t zero = 0;
    check(dbhip_sort_perm(&key, &zero, &zero, 1, n, 0, (uint32_t*)perm->ptr(), nullptr));
  }
  int64_t at = 0;
  for (uint64_t c : counts) {
    out.push_back(take_block(block, (const uint32_t*)perm->ptr() + at, (int64_t)c));
    at += (int64_t)c;
  }
  return out;
}
// HashFlightScatter / OneHashKeyFlightScatter (flight_scatter_hash.rs:57-330): siphash64(key) % scatter_size, several keys through a
// DefaultHasher, NULL keys to the default scatter index — dbhip_scatter_indices computes the reference's own values
class HashFlightScatter : public FlightScatter {
 public:
  HashFlightScatter(std::vector<size_t> hash_keys, size_t scatter_size, uint64_t default_scatter_index = 0)
      : keys_(std::move(hash_keys)), scatter_size_(scatter_size), default_(default_scatter_index) {}
  const char* name() const override { return keys_.size() == 1 ? "OneHashKey" : "Hash"; }
  // the destination of every row + rows per destination (FlightScatter::scatter_indices)
  std::pair<Buf, std::vector<uint64_t>> scatter_indices(const DataBlock& b) const {
    std::vector<dbhip_col> cols;
    for (size_t k : keys_) cols.push_back(b.get_by_offset(k).c());
    Buf idx = make_buf((size_t)(b.num_rows > 0 ? b.num_rows : 1) * 4 + 64), cnt = make_buf(scatter_size_ * 8);
    check(dbhip_scatter_indices(cols.data(), (int32_t)cols.size(), b.num_rows, (uint32_t)scatter_size_, default_, (uint32_t*)idx->ptr(),
                                (uint64_t*)cnt->ptr(), nullptr));
    std::vector<uint64_t> counts(scatter_size_);
    cnt->download(counts.data(), scatter_size_ * 8);
    return {idx, counts};
  }
  std::vector<DataBlock> execute(const DataBlock& data_block) override {
    auto ic = scatter_indices(data_block);
    return scatter_block(data_block, ic.first, ic.second);
  }
 private:
  std::vector<size_t> keys_;
  size_t scatter_size_;
  uint64_t default_;
};
// SortBoundScatter over ordered bounds (sort_exchange_injector.rs + sort_spill.rs:1008-1040): range i = rows after bound i - 1 up
// to and including bound i, to destination i % scatter_size is the caller's choice; here one block per range
class SortBoundScatter : public FlightScatter {
 public:
  SortBoundScatter(std::vector<SortColumnDescription> desc, DataBlock bounds) : desc_(std::move(desc)), bounds_(std::move(bounds)) {}
  const char* name() const override { return "SortBound"; }
  std::vector<DataBlock> execute(const DataBlock& b) override {
    std::vector<dbhip_col> keys, bnd;
    std::vector<uint8_t> d, nf;
    for (size_t i = 0; i < desc_.size(); ++i) {
      keys.push_back(b.get_by_offset(desc_[i].offset).c());
      if (bounds_.num_rows) bnd.push_back(bounds_.get_by_offset(i).c());
      d.push_back(desc_[i].asc ? 0 : 1); nf.push_back(desc_[i].nulls_first ? 1 : 0);
    }
    const int64_t nb = bounds_.num_rows;
    Buf part = make_buf((size_t)(b.num_rows > 0 ? b.num_rows : 1) * 4 + 64), cnt = make_buf((size_t)(nb + 1) * 8);
    check(dbhip_sort_bound_partition(keys.data(), nb ? bnd.data() : nullptr, d.data(), nf.data(), (int32_t)keys.size(), b.num_rows, nb,
                                     (uint32_t*)part->ptr(), (uint64_t*)cnt->ptr(), nullptr));
    std::vector<uint64_t> counts((size_t)nb + 1);
    cnt->download(counts.data(), counts.size() * 8);
    return scatter_block(b, part, counts);
  }
 private:
  std::vector<SortColumnDescription> desc_;
  DataBlock bounds_;   // column i = the bound values of sort key i, ordered by the same keys
};



// ---- scan side: one leaf column of column_chunks_to_record_batch --------------------------------
// (fuse/src/io/read/block/parquet/deserialize.rs:33-81 + the arrow -> Column conversion). `bytes` is the column chunk as the
// block reader fetched it (DataItem::RawData). std::nullopt = the library declines the chunk (compressed, nested, DELTA_*
// encodings): the caller keeps the arrow-rs reader for it, exactly like a ScalarFunction falls back on DBHIP_ERR_UNSUPPORTED.
struct ParquetLeaf {        // what ColumnDescriptor + ColumnChunkMetaData give
  int32_t physical_type;    // parquet.thrift Type
  int32_t type_length = 0;  // FIXED_LEN_BYTE_ARRAY
  int32_t max_def_level = 0, max_rep_level = 0;
  int32_t codec = 0;        // parquet.thrift CompressionCodec (0 = UNCOMPRESSED)
};

inline std::optional<Column> column_chunk_to_column(const uint8_t* bytes, size_t len, const ParquetLeaf& leaf, DataType field_type) {
  dbhip_pq_chunk* h = nullptr;
  dbhip_pq_info info;
  const int32_t rc = dbhip_pq_chunk_open(bytes, (int64_t)len, leaf.codec, leaf.physical_type, leaf.type_length, leaf.max_def_level,
                                         leaf.max_rep_level, field_type.id, &h, &info);
  if (rc == DBHIP_ERR_UNSUPPORTED) return std::nullopt;
  check(rc);
  struct Closer { dbhip_pq_chunk* h; ~Closer() { dbhip_pq_chunk_close(h); } } closer{h};
  Column c;
  c.type = field_type;
  c.type.nullable = info.has_validity != 0;
  c.len = info.num_values;
  // resident copy (+ slack) of what the device decodes — the chunk, or the decompressed page stream of a compressed chunk;
  // buffer 0 of a String column
  const uint8_t* img = nullptr;
  int64_t img_len = 0;
  check(dbhip_pq_chunk_image(h, &img, &img_len));
  if (!img) { img = bytes; img_len = (int64_t)len; }
  Buf chunk = make_buf((size_t)img_len + 8);
  chunk->upload(img, (size_t)img_len);
  c.data = make_buf((size_t)info.out_bytes);
  if (info.has_validity) c.validity = make_buf((size_t)info.validity_bytes);
  check(dbhip_pq_chunk_decode(h, (const uint8_t*)chunk->ptr(), c.data->ptr(), c.validity ? (uint8_t*)c.validity->ptr() : nullptr, nullptr));
  check(dbhip_stream_sync(nullptr));
  if (field_type.id == DBHIP_T_STRING) {
    c.str_data = chunk;
    void* p = chunk->ptr();
    c.str_ptrs = make_buf(sizeof(void*));
    c.str_ptrs->upload(&p, sizeof(void*));
  }
  return c;
}

// ---- the same for a whole block, device mode (round 5): every leaf's page headers are read on the host, the chunks go to HBM AS
// STORED, and ONE dbhip_pq_chunks_decode_device call decompresses (ZSTD / LZ4 / Snappy) and decodes all of them. A leaf the library
// declines (std::nullopt in the result) stays with the arrow-rs reader; a malformed chunk raises like any decode error.
struct ChunkBytes { const uint8_t* bytes; size_t len; };
inline std::vector<std::optional<Column>> column_chunks_to_columns(const std::vector<ChunkBytes>& chunks, const std::vector<ParquetLeaf>& leaves,
                                                                   const std::vector<DataType>& field_types) {
  const size_t n = chunks.size();
  std::vector<std::optional<Column>> out(n);
  std::vector<dbhip_pq_chunk*> h;
  std::vector<size_t> which;
  std::vector<dbhip_pq_info> info;
  struct Closer { std::vector<dbhip_pq_chunk*>& v; ~Closer() { for (auto* c : v) dbhip_pq_chunk_close(c); } } closer{h};
  for (size_t i = 0; i < n; ++i) {
    dbhip_pq_chunk* c = nullptr;
    dbhip_pq_info f;
    const int32_t rc = dbhip_pq_chunk_open_device(chunks[i].bytes, (int64_t)chunks[i].len, leaves[i].codec, leaves[i].physical_type, leaves[i].type_length,
                                                  leaves[i].max_def_level, leaves[i].max_rep_level, field_types[i].id, &c, &f);
    if (rc == DBHIP_ERR_UNSUPPORTED) continue;
    check(rc);
    h.push_back(c); which.push_back(i); info.push_back(f);
  }
  const size_t m = h.size();
  if (m == 0) return out;
  std::vector<Buf> chunk_dev(m), image_dev(m);
  std::vector<const uint8_t*> cd(m);
  std::vector<uint8_t*> im(m), val(m);
  std::vector<void*> ov(m);
  std::vector<Column> cols(m);
  for (size_t k = 0; k < m; ++k) {
    const size_t i = which[k];
    chunk_dev[k] = make_buf(chunks[i].len + 32);          // (16-byte aligned allocation, readable past the end to a multiple of 16)
    chunk_dev[k]->upload(chunks[i].bytes, chunks[i].len);
    if (info[k].image_bytes) image_dev[k] = make_buf((size_t)info[k].image_bytes);
    Column& c = cols[k];
    c.type = field_types[i];
    c.type.nullable = info[k].has_validity != 0;
    c.len = info[k].num_values;
    c.data = make_buf((size_t)info[k].out_bytes + 16);
    if (info[k].has_validity) c.validity = make_buf((size_t)info[k].validity_bytes + 8);
    cd[k] = (const uint8_t*)chunk_dev[k]->ptr();
    im[k] = image_dev[k] ? (uint8_t*)image_dev[k]->ptr() : nullptr;
    ov[k] = c.data->ptr();
    val[k] = c.validity ? (uint8_t*)c.validity->ptr() : nullptr;
  }
  std::vector<int64_t> nulls(m);
  std::vector<int32_t> status(m);
  const int32_t rc = dbhip_pq_chunks_decode_device(h.data(), (int32_t)m, cd.data(), im.data(), ov.data(), val.data(), nulls.data(), status.data(), nullptr);
  for (size_t k = 0; k < m; ++k) {
    if (status[k] == DBHIP_ERR_UNSUPPORTED) continue;     // (a form found on the device that stays with the CPU reader)
    if (status[k] != DBHIP_OK) check(rc);
    Column& c = cols[k];
    if (field_types[which[k]].id == DBHIP_T_STRING) {      // views point into the decompressed image (or the chunk itself)
      c.str_data = image_dev[k] ? image_dev[k] : chunk_dev[k];
      void* p = c.str_data->ptr();
      c.str_ptrs = make_buf(sizeof(void*));
      c.str_ptrs->upload(&p, sizeof(void*));
    }
    out[which[k]] = std::move(c);
  }
  return out;
}

// Array(T) of a primitive leaf (one repeated ancestor): -> { offsets [rows + 1] (host), NULL lists (host, empty when the list is
// required), the element column }; std::nullopt when the library declines the chunk
struct ListColumn {
  std::vector<uint64_t> offsets;
  std::vector<uint8_t> list_valid;   // one byte per row, empty = no NULL lists possible
  Column elements;
};
inline std::optional<ListColumn> list_chunk_to_column(const uint8_t* bytes, size_t len, const ParquetLeaf& leaf, bool list_nullable, bool element_nullable,
                                                      DataType element_type) {
  dbhip_pq_chunk* h = nullptr;
  dbhip_pq_info info;
  const int32_t rc = dbhip_pq_chunk_open_device_list(bytes, (int64_t)len, leaf.codec, leaf.physical_type, leaf.type_length, list_nullable ? 1 : 0,
                                                     element_nullable ? 1 : 0, element_type.id, &h, &info);
  if (rc == DBHIP_ERR_UNSUPPORTED) return std::nullopt;
  check(rc);
  struct Closer { dbhip_pq_chunk* h; ~Closer() { dbhip_pq_chunk_close(h); } } closer{h};
  Buf chunk = make_buf(len + 32), image = info.image_bytes ? make_buf((size_t)info.image_bytes) : nullptr;
  chunk->upload(bytes, len);
  Buf offs = make_buf((size_t)(info.num_values + 1) * 8 + 16), lval = list_nullable ? make_buf((size_t)info.validity_bytes + 8) : nullptr;
  ListColumn L;
  Column& c = L.elements;
  c.type = element_type;
  c.type.nullable = element_nullable;
  c.data = make_buf((size_t)info.out_bytes + 16);
  if (element_nullable) c.validity = make_buf((size_t)info.validity_bytes + 8);
  int64_t rows = 0, elems = 0, null_lists = 0;
  check(dbhip_pq_chunk_decode_device_list(h, (const uint8_t*)chunk->ptr(), image ? (uint8_t*)image->ptr() : nullptr, (uint64_t*)offs->ptr(),
                                          lval ? (uint8_t*)lval->ptr() : nullptr, c.data->ptr(), c.validity ? (uint8_t*)c.validity->ptr() : nullptr,
                                          &rows, &elems, &null_lists, nullptr));
  c.len = elems;
  L.offsets.resize((size_t)rows + 1);
  offs->download(L.offsets.data(), L.offsets.size() * 8);
  if (lval) {
    std::vector<uint8_t> bits((size_t)(rows + 7) / 8);
    if (!bits.empty()) lval->download(bits.data(), bits.size());
    L.list_valid.resize((size_t)rows);
    for (int64_t r = 0; r < rows; ++r) L.list_valid[(size_t)r] = (bits[(size_t)r >> 3] >> (r & 7)) & 1;
  }
  if (element_type.id == DBHIP_T_STRING) {
    c.str_data = image ? image : chunk;
    void* p = c.str_data->ptr();
    c.str_ptrs = make_buf(sizeof(void*));
    c.str_ptrs->upload(&p, sizeof(void*));
  }
  return L;
}

}  // namespace dbhip_host
