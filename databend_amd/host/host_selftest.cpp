// host_selftest.cpp — drives the C++ host mirror (dbhip_host.hpp) the way the reference's pipeline
// would, on one MI355X, and checks every result against plain host loops written here
// (closed forms / scalar C++; independent of oracle/). Exit code 0 = all checks passed.
//   1. BASELINE configs[0] twin: SELECT sum(a + b * c) over Int64 columns through Evaluator +
//      the single-state aggregator (wrapping arithmetic)
//   2. row errors: divide by zero -> "divided by zero while evaluating function `divide(..)` in expr `..`",
//      first failing row, NULL rows never raise, selection honoured
//   3. and_filters + FilterExecutor
//   4. TPC-H Q1 as the reference's plan: TransformFilter -> CompoundBlockOperator (decimal maps) ->
//      TransformPartialAggregate x2 -> TransformFinalAggregate, against an __int128 host loop
//   5. InnerHashJoin (Join / JoinStream) and DataBlock::sort
//   6. cosine_distance via the registry
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <random>
#include <set>
#include <thread>
#include <tuple>

#include "dbhip_host.hpp"

using namespace dbhip_host;

static int g_fail = 0;
#define CHECK(cond)                                                        \
  do {                                                                     \
    if (!(cond)) { printf("CHECK FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond); ++g_fail; } \
  } while (0)

static void test_sum_a_plus_b_mul_c() {
  const int64_t n = 1000003;
  std::mt19937_64 rng(1);
  std::vector<int64_t> a(n), b(n), c(n);
  for (int64_t i = 0; i < n; ++i) { a[i] = (int64_t)rng(); b[i] = (int64_t)rng(); c[i] = (int64_t)rng(); }  // full range: wraps
  DataBlock block({Column::from_vector(DataType::of(DBHIP_T_I64), a), Column::from_vector(DataType::of(DBHIP_T_I64), b),
                   Column::from_vector(DataType::of(DBHIP_T_I64), c)}, n);
  auto I64 = DataType::of(DBHIP_T_I64);
  Expr e = Expr::call("plus", {Expr::column_ref(0, I64, "a"), Expr::call("multiply", {Expr::column_ref(1, I64, "b"), Expr::column_ref(2, I64, "c")})});
  CHECK(e.sql_display() == "(a + (b * c))");
  CHECK(e.data_type().id == DBHIP_T_I64);
  Evaluator ev(block);
  Value v = ev.run(e);
  CHECK(!v.is_scalar && v.column.len == n);
  std::vector<int64_t> got = v.column.to_vector<int64_t>();
  uint64_t exp_sum = 0;
  bool all = true;
  for (int64_t i = 0; i < n; ++i) {
    uint64_t x = (uint64_t)a[i] + (uint64_t)b[i] * (uint64_t)c[i];
    all &= (uint64_t)got[i] == x;
    exp_sum += x;
  }
  CHECK(all);
  // the same tree as ONE launch (Evaluator::run_fused -> dbhip_expr_eval) == the node-by-node result
  auto fused = ev.run_fused(e);
  CHECK(fused.has_value() && fused->column.len == n && fused->column.to_vector<int64_t>() == got);
  {  // mixed widths, a nullable input and a comparison on top: (a32 * u8 - 7) >= b64   ->  Boolean NULL
    std::vector<int32_t> a32(n); std::vector<uint8_t> u8v(n); std::vector<bool> valid(n);
    for (int64_t i = 0; i < n; ++i) { a32[i] = (int32_t)rng(); u8v[i] = (uint8_t)rng(); valid[i] = (rng() & 7) != 0; }
    DataBlock blk({Column::from_vector(DataType::of(DBHIP_T_I32), a32, &valid), Column::from_vector(DataType::of(DBHIP_T_U8), u8v),
                   Column::from_vector(I64, b)}, n);
    Expr lhs = Expr::call("minus", {Expr::call("multiply", {Expr::column_ref(0, DataType::of(DBHIP_T_I32, true), "x"), Expr::column_ref(1, DataType::of(DBHIP_T_U8), "y")}),
                                    Expr::constant(Scalar::Int(DBHIP_T_U8, 7))});
    Expr pred = Expr::call("gte", {lhs, Expr::column_ref(2, I64, "b")});
    Evaluator ev2(blk);
    Value stepwise = ev2.run(pred);
    auto f2 = ev2.run_fused(pred);
    CHECK(f2.has_value());
    if (f2) {
      CHECK(f2->column.to_bools() == stepwise.column.to_bools());
      CHECK(f2->column.validity_to_host() == valid);
    }
    // a decimal node is outside the fused subset: the caller keeps the node-by-node path
    Expr dec = Expr::call("plus", {Expr::constant(Scalar::Dec(15, 2, 100)), Expr::constant(Scalar::Dec(15, 2, 1))});
    CHECK(!ev2.run_fused(dec).has_value());
  }
  DataBlock mapped({v.column}, n);
  SingleStateAggregator agg({{"sum", 0, I64}, {"count", std::nullopt, DataType()}});
  agg.transform(mapped);
  agg.transform(mapped);  // two blocks
  CHECK((uint64_t)agg.int_result(0) == exp_sum * 2);
  CHECK(agg.int_result(1) == 2 * n);
  // mixed widths follow ResultTypeOfBinary: Int32 + Int32 -> Int64, UInt8 * Int8 -> Int16
  CHECK(Expr::call("plus", {Expr::column_ref(0, DataType::of(DBHIP_T_I32), "x"), Expr::column_ref(1, DataType::of(DBHIP_T_I32), "y")}).data_type().id == DBHIP_T_I64);
  CHECK(Expr::call("multiply", {Expr::column_ref(0, DataType::of(DBHIP_T_U8), "x"), Expr::column_ref(1, DataType::of(DBHIP_T_I8), "y")}).data_type().id == DBHIP_T_I16);
}

static void test_row_errors() {
  std::vector<int64_t> a = {10, 20, 30, 40, 50}, b = {2, 0, 5, 0, 1};
  std::vector<bool> bvalid = {true, false, true, true, true};  // row 1 is NULL: never raises
  auto I64 = DataType::of(DBHIP_T_I64);
  DataBlock block({Column::from_vector(I64, a), Column::from_vector(I64, b, &bvalid)}, 5);
  Expr e = Expr::call("divide", {Expr::column_ref(0, I64, "a"), Expr::column_ref(1, I64.wrap_nullable(), "b")});
  CHECK(e.data_type().id == DBHIP_T_F64 && e.data_type().nullable);
  Evaluator ev(block);
  bool thrown = false;
  try { ev.run(e); } catch (const ErrorCode& err) {
    thrown = true;
    CHECK(err.kind == "BadArguments");
    CHECK(std::string(err.what()) == "divided by zero while evaluating function `divide(40, 0)` in expr `(a / b)`");
  }
  CHECK(thrown);
  // with a selection that excludes the failing row the expression evaluates (render_error honours it)
  std::vector<uint32_t> sel = {0, 2, 4};
  Value v = ev.run_with_selection(e, &sel);
  std::vector<double> q = v.column.to_vector<double>();
  CHECK(q[0] == 5.0 && q[2] == 6.0 && q[4] == 50.0);
  std::vector<bool> valid = v.column.validity_to_host();
  CHECK(valid[0] && !valid[1] && valid[2]);
  // no error at all when the divisor is never zero on valid rows
  std::vector<int64_t> b2 = {2, 0, 5, 4, 1};
  DataBlock ok({Column::from_vector(I64, a), Column::from_vector(I64, b2, &bvalid)}, 5);
  Evaluator ev2(ok);
  CHECK(ev2.run(e).column.to_vector<double>()[3] == 10.0);
}

static void test_filter() {
  const int64_t n = 100000;
  std::mt19937 rng(3);
  std::vector<int32_t> d(n); std::vector<int64_t> x(n);
  for (int64_t i = 0; i < n; ++i) { d[i] = (int32_t)(rng() % 1000); x[i] = (int64_t)(rng() % 100) - 50; }
  auto DATE = DataType::of(DBHIP_T_DATE); auto I64 = DataType::of(DBHIP_T_I64);
  DataBlock block({Column::from_vector(DATE, d), Column::from_vector(I64, x)}, n);
  Expr pred = Expr::call("and_filters", {Expr::call("lte", {Expr::column_ref(0, DATE, "d"), Expr::constant(Scalar::Int(DBHIP_T_DATE, 700))}),
                                         Expr::call("gt", {Expr::column_ref(1, I64, "x"), Expr::constant(Scalar::Int(DBHIP_T_I64, 0))})});
  TransformFilter tf(pred);
  DataBlock out = tf.transform(block);
  std::vector<int32_t> od = out.columns[0].to_vector<int32_t>(); std::vector<int64_t> ox = out.columns[1].to_vector<int64_t>();
  std::vector<int32_t> ed; std::vector<int64_t> ex;
  for (int64_t i = 0; i < n; ++i) if (d[i] <= 700 && x[i] > 0) { ed.push_back(d[i]); ex.push_back(x[i]); }
  CHECK(out.num_rows == (int64_t)ed.size() && od == ed && ox == ex);
}

static void test_or_filters() {
  // or_filters / and_filters over NULLABLE predicates (evaluator.rs:1802-1880): NULL decodes to FALSE, the result is never NULL;
  // node by node (run) and as one fused program (run_fused) — and a later argument never raises on rows an earlier one took
  const int64_t n = 50001;
  std::mt19937 rng(5);
  std::vector<int64_t> a(n), b(n); std::vector<bool> va(n), vb(n);
  for (int64_t i = 0; i < n; ++i) { a[i] = (int64_t)(rng() % 10); b[i] = (int64_t)(rng() % 10); va[i] = rng() % 4 != 0; vb[i] = rng() % 3 != 0; }
  auto I64 = DataType::of(DBHIP_T_I64);
  DataBlock block({Column::from_vector(I64, a, &va), Column::from_vector(I64, b, &vb)}, n);
  Expr pa = Expr::call("gt", {Expr::column_ref(0, I64.wrap_nullable(), "a"), Expr::constant(Scalar::Int(DBHIP_T_I64, 3))});
  Expr pb = Expr::call("lt", {Expr::column_ref(1, I64.wrap_nullable(), "b"), Expr::constant(Scalar::Int(DBHIP_T_I64, 8))});
  Evaluator ev(block);
  for (int is_or = 0; is_or < 2; ++is_or) {
    Expr e = Expr::call(is_or ? "or_filters" : "and_filters", {pa, pb});
    std::vector<bool> exp(n);
    for (int64_t i = 0; i < n; ++i) { bool x = va[i] && a[i] > 3, y = vb[i] && b[i] < 8; exp[i] = is_or ? (x || y) : (x && y); }
    Value v = ev.run(e);
    CHECK(!v.column.validity);
    CHECK(Column::unpack_bits(v.column.data, n) == exp);
    auto f = ev.run_fused(e);
    CHECK(f.has_value());
    if (f->column.validity) { std::vector<bool> vv = Column::unpack_bits(f->column.validity, n); for (int64_t i = 0; i < n; ++i) CHECK(vv[(size_t)i]); }
    CHECK(Column::unpack_bits(f->column.data, n) == exp);
  }
  // or_filters(b = 0, 10 / b > 2): the division only sees the rows where b != 0
  std::vector<int64_t> d(n);
  for (int64_t i = 0; i < n; ++i) d[i] = (int64_t)(rng() % 5);
  DataBlock blk2({Column::from_vector(I64, d)}, n);
  Expr isz = Expr::call("eq", {Expr::column_ref(0, I64, "d"), Expr::constant(Scalar::Int(DBHIP_T_I64, 0))});
  Expr quo = Expr::call("gt", {Expr::call("divide", {Expr::constant(Scalar::Int(DBHIP_T_I64, 10)), Expr::column_ref(0, I64, "d")}),
                               Expr::constant(Scalar::Float(DBHIP_T_F64, 2.0))});
  Value r = Evaluator(blk2).run(Expr::call("or_filters", {isz, quo}));
  std::vector<bool> got = Column::unpack_bits(r.column.data, n);
  for (int64_t i = 0; i < n; ++i) CHECK(got[(size_t)i] == (d[i] == 0 || 10.0 / (double)d[i] > 2.0));
}

static void test_selector() {
  // FilterExecutor through the Selector (true / false lists, short-circuit And / Or) == through the predicate's Bitmap, on a tree
  // that mixes both, nullable operands and a Boolean column; a tree with arithmetic under a comparison falls back by itself
  const int64_t n = 200003;
  std::mt19937 rng(23);
  std::vector<int64_t> a(n), b(n); std::vector<double> c(n); std::vector<bool> d(n), va(n);
  for (int64_t i = 0; i < n; ++i) { a[i] = (int64_t)(rng() % 100) - 50; b[i] = (int64_t)(rng() % 100) - 50; c[i] = (double)(rng() % 1000) / 1000.0; d[i] = rng() % 2; va[i] = rng() % 5 != 0; }
  auto I64 = DataType::of(DBHIP_T_I64); auto F64 = DataType::of(DBHIP_T_F64); auto B = DataType::of(DBHIP_T_BOOL);
  DataBlock block({Column::from_vector(I64, a, &va), Column::from_vector(I64, b), Column::from_vector(F64, c), Column::from_bools(d)}, n);
  auto col = [&](size_t i, DataType t, const char* nm) { return Expr::column_ref(i, t, nm); };
  Expr p1 = Expr::call("gt", {col(0, I64.wrap_nullable(), "a"), Expr::constant(Scalar::Int(DBHIP_T_I64, 10))});
  Expr p2 = Expr::call("lte", {col(1, I64, "b"), Expr::constant(Scalar::Int(DBHIP_T_I64, 0))});
  Expr p3 = Expr::call("lt", {col(2, F64, "c"), Expr::constant(Scalar::Float(DBHIP_T_F64, 0.5))});
  Expr p4 = Expr::call("noteq", {col(0, I64.wrap_nullable(), "a"), col(1, I64, "b")});
  Expr tree = Expr::call("and_filters", {p3, Expr::call("or_filters", {p1, col(3, B, "d"), p2}), p4});
  std::optional<Selection> s = Selector(block).select(tree);
  CHECK(s.has_value());
  std::vector<int64_t> exp;
  for (int64_t i = 0; i < n; ++i) if (c[i] < 0.5 && ((va[i] && a[i] > 10) || d[i] || b[i] <= 0) && (va[i] && a[i] != b[i])) exp.push_back(b[i] * 1000 + i % 1000);
  FilterExecutor fe(tree);
  DataBlock out = fe.filter_with_selector(block), out2 = fe.filter_with_bitmap(block);
  CHECK(out.num_rows == (int64_t)exp.size() && out2.num_rows == out.num_rows);
  // an And over ascending lists keeps row order only inside each Or branch: compare as multisets with the Bitmap path, and the And-only
  // tree in exact order
  auto key = [](const DataBlock& blk) { std::multiset<int64_t> m; auto bb = blk.columns[1].to_vector<int64_t>(); auto cc = blk.columns[2].to_vector<double>(); for (size_t i = 0; i < bb.size(); ++i) m.insert(bb[i] * 100000 + (int64_t)(cc[i] * 1000)); return m; };
  CHECK(key(out) == key(out2));
  Expr conj = Expr::call("and_filters", {p3, p2, p4});
  DataBlock o3 = FilterExecutor(conj).filter_with_selector(block), o4 = FilterExecutor(conj).filter_with_bitmap(block);
  CHECK(o3.num_rows == o4.num_rows && o3.columns[1].to_vector<int64_t>() == o4.columns[1].to_vector<int64_t>() && o3.columns[2].to_vector<double>() == o4.columns[2].to_vector<double>());
  // arithmetic under a comparison: not a Selector leaf -> the Bitmap path, same API
  Expr other = Expr::call("gt", {Expr::call("plus", {col(1, I64, "b"), col(1, I64, "b")}), Expr::constant(Scalar::Int(DBHIP_T_I64, 10))});
  CHECK(!Selector(block).select(other).has_value());
  int64_t cnt = 0;
  for (int64_t i = 0; i < n; ++i) cnt += 2 * b[i] > 10;
  CHECK(FilterExecutor(other).filter_with_selector(block).num_rows == cnt && FilterExecutor(other).filter(block).num_rows == cnt);
}

static void test_q1_plan() {
  const int64_t n = 300007;
  std::mt19937_64 rng(2);
  std::vector<int64_t> qty(n), price(n), disc(n), tax(n); std::vector<int32_t> ship(n); std::vector<std::string> rf(n), ls(n);
  const int32_t cutoff = 10471;
  for (int64_t i = 0; i < n; ++i) {
    qty[i] = (int64_t)(rng() % 50 + 1) * 100; price[i] = 90000 + (int64_t)(rng() % 10404951); disc[i] = (int64_t)(rng() % 11); tax[i] = (int64_t)(rng() % 9);
    ship[i] = 8036 + (int32_t)(rng() % 2526);
    rf[i] = std::string(1, "ANR"[rng() % 3]); ls[i] = std::string(1, "FO"[rng() % 2]);
  }
  auto D152 = DataType::Decimal(15, 2); auto DATE = DataType::of(DBHIP_T_DATE); auto STR = DataType::of(DBHIP_T_STRING);
  // columns: 0 qty 1 price 2 disc 3 tax 4 returnflag 5 linestatus 6 shipdate
  auto make_block = [&](int64_t lo, int64_t hi) {
    auto sl = [&](const std::vector<int64_t>& v) { return std::vector<int64_t>(v.begin() + lo, v.begin() + hi); };
    return DataBlock({Column::from_vector(D152, sl(qty)), Column::from_vector(D152, sl(price)), Column::from_vector(D152, sl(disc)),
                      Column::from_vector(D152, sl(tax)), Column::from_short_strings(std::vector<std::string>(rf.begin() + lo, rf.begin() + hi)),
                      Column::from_short_strings(std::vector<std::string>(ls.begin() + lo, ls.begin() + hi)),
                      Column::from_vector(DATE, std::vector<int32_t>(ship.begin() + lo, ship.begin() + hi))}, hi - lo);
  };
  // plan (benchmark/tpch/queries/01.sql after the planner's rewrites)
  TransformFilter filter(Expr::call("lte", {Expr::column_ref(6, DATE, "l_shipdate"), Expr::constant(Scalar::Int(DBHIP_T_DATE, cutoff))}));
  Expr one = Expr::constant(Scalar::Int(DBHIP_T_U8, 1));
  Expr one_minus = Expr::call("minus", {one, Expr::column_ref(2, D152, "l_discount")});
  Expr disc_price = Expr::call("multiply", {Expr::column_ref(1, D152, "l_extendedprice"), one_minus});
  Expr charge = Expr::call("multiply", {disc_price, Expr::call("plus", {one, Expr::column_ref(3, D152, "l_tax")})});
  CHECK(disc_price.data_type().precision == 31 && disc_price.data_type().scale == 4 && disc_price.data_type().id == DBHIP_T_DEC128);
  CHECK(charge.data_type().precision == 38 && charge.data_type().scale == 6);
  TransformMap map({disc_price, charge});  // appended as columns 7, 8
  AggregatorParams params;
  params.group_columns = {4, 5}; params.group_data_types = {STR, STR};
  params.aggregate_functions = {{"sum", 0, D152}, {"sum", 1, D152}, {"sum", 7, disc_price.data_type()}, {"sum", 8, charge.data_type()}, {"sum", 2, D152}, {"count", std::nullopt, DataType()}};
  TransformPartialAggregate partial_a(params), partial_b(params);  // two pipeline lanes
  TransformFinalAggregate final_(params);
  const int64_t BLOCK = 65536;
  int lane = 0;
  for (int64_t lo = 0; lo < n; lo += BLOCK, ++lane) {
    DataBlock b = map.transform(filter.transform(make_block(lo, std::min(n, lo + BLOCK))));
    (lane % 2 ? partial_b : partial_a).transform(std::move(b));
  }
  for (auto* p : {&partial_a, &partial_b})
    for (auto& meta : p->on_finish(true)) final_.transform(std::move(meta));
  std::vector<DataBlock> res = final_.on_finish(true);
  CHECK(res.size() == 1);
  const DataBlock& r = res[0];
  // expected
  struct G { int64_t q = 0, p = 0, d = 0; __int128 dp = 0, ch = 0; uint64_t c = 0; };
  std::map<std::pair<std::string, std::string>, G> exp;
  for (int64_t i = 0; i < n; ++i) if (ship[i] <= cutoff) {
    G& g = exp[{rf[i], ls[i]}];
    __int128 dp = (__int128)price[i] * (100 - disc[i]);
    g.q += qty[i]; g.p += price[i]; g.d += disc[i]; g.dp += dp; g.ch += dp * (100 + tax[i]); g.c += 1;
  }
  CHECK(r.num_rows == (int64_t)exp.size());
  // result block = [aggregate results..., group columns...]
  auto sq = r.columns[0].to_vector<int64_t>(); auto sp = r.columns[1].to_vector<int64_t>();
  auto sdp = r.columns[2].to_vector<__int128>(); auto sch = r.columns[3].to_vector<__int128>();
  auto sd = r.columns[4].to_vector<int64_t>(); auto cnt = r.columns[5].to_vector<uint64_t>();
  auto krf = r.columns[6].to_short_strings(); auto kls = r.columns[7].to_short_strings();
  CHECK(r.columns[2].type.precision == 38 && r.columns[2].type.scale == 4 && r.columns[0].type.precision == 18);
  for (int64_t i = 0; i < r.num_rows; ++i) {
    auto it = exp.find({krf[(size_t)i], kls[(size_t)i]});
    CHECK(it != exp.end());
    if (it == exp.end()) continue;
    const G& g = it->second;
    CHECK(sq[i] == g.q && sp[i] == g.p && sd[i] == g.d && sdp[i] == g.dp && sch[i] == g.ch && cnt[i] == g.c);
  }
  // ORDER BY l_returnflag, l_linestatus on the result block
  DataBlock sorted = sort_block(r, {{6, true, false}, {7, true, false}});
  auto s1 = sorted.columns[6].to_short_strings(); auto s2 = sorted.columns[7].to_short_strings();
  std::vector<std::pair<std::string, std::string>> got_keys, exp_keys;
  for (size_t i = 0; i < s1.size(); ++i) got_keys.push_back({s1[i], s2[i]});
  for (auto& kv : exp) exp_keys.push_back(kv.first);
  CHECK(got_keys == exp_keys);
}

// Round 6: the drop-in at the reference's block size. Q1 fed as 65,536-row... here 20,000-row blocks through
// TransformFusedPartialAggregate (pipelined table, multi-block launches, checkpoint at on_finish) from two pipeline lanes, against
// the three-operator plan of test_q1_plan's shape and an __int128 host loop; then a stream whose 12th block holds 60 groups: the
// checkpoint gives the uncommitted blocks back and the operator replays them through TransformFilter -> maps -> add_groups;
// and BlockAccumulator: squashed blocks == the concatenation.
static void test_fused_partial_aggregate() {
  const int64_t n = 400003, BLOCK = 20000;
  std::mt19937_64 rng(12);
  std::vector<int64_t> qty(n), price(n), disc(n), tax(n); std::vector<int32_t> ship(n); std::vector<std::string> rf(n), ls(n);
  const int32_t cutoff = 10471;
  for (int64_t i = 0; i < n; ++i) {
    qty[i] = (int64_t)(rng() % 50 + 1) * 100; price[i] = 90000 + (int64_t)(rng() % 10404951); disc[i] = (int64_t)(rng() % 11); tax[i] = (int64_t)(rng() % 9);
    ship[i] = 8036 + (int32_t)(rng() % 2526);
    rf[i] = std::string(1, "ANR"[rng() % 3]); ls[i] = std::string(1, "FO"[rng() % 2]);
  }
  auto D152 = DataType::Decimal(15, 2); auto DATE = DataType::of(DBHIP_T_DATE); auto STR = DataType::of(DBHIP_T_STRING);
  auto make_block = [&](int64_t lo, int64_t hi) {
    auto sl = [&](const std::vector<int64_t>& v) { return std::vector<int64_t>(v.begin() + lo, v.begin() + hi); };
    return DataBlock({Column::from_vector(D152, sl(qty)), Column::from_vector(D152, sl(price)), Column::from_vector(D152, sl(disc)),
                      Column::from_vector(D152, sl(tax)), Column::from_short_strings(std::vector<std::string>(rf.begin() + lo, rf.begin() + hi)),
                      Column::from_short_strings(std::vector<std::string>(ls.begin() + lo, ls.begin() + hi)),
                      Column::from_vector(DATE, std::vector<int32_t>(ship.begin() + lo, ship.begin() + hi))}, hi - lo);
  };
  Expr pred = Expr::call("lte", {Expr::column_ref(6, DATE, "l_shipdate"), Expr::constant(Scalar::Int(DBHIP_T_DATE, cutoff))});
  Expr one = Expr::constant(Scalar::Int(DBHIP_T_U8, 1));
  Expr disc_price = Expr::call("multiply", {Expr::column_ref(1, D152, "l_extendedprice"), Expr::call("minus", {one, Expr::column_ref(2, D152, "l_discount")})});
  Expr charge = Expr::call("multiply", {disc_price, Expr::call("plus", {one, Expr::column_ref(3, D152, "l_tax")})});
  AggregatorParams params;
  params.group_columns = {4, 5}; params.group_data_types = {STR, STR};
  params.aggregate_functions = {{"sum", 0, D152}, {"sum", 1, D152}, {"sum", std::nullopt, disc_price.data_type()}, {"sum", std::nullopt, charge.data_type()},
                                {"sum", 2, D152}, {"count", std::nullopt, DataType()}};
  params.aggregate_functions[2].arg = 0; params.aggregate_functions[3].arg = 0;   // (has an argument; the fused operator takes it from `args`)
  std::vector<std::optional<Expr>> args = {Expr::column_ref(0, D152, "l_quantity"), Expr::column_ref(1, D152, "l_extendedprice"), disc_price, charge,
                                           Expr::column_ref(2, D152, "l_discount"), std::nullopt};
  void* s1 = nullptr; void* s2 = nullptr;
  check(dbhip_stream_create(&s1)); check(dbhip_stream_create(&s2));
  {
    TransformFusedPartialAggregate lane_a(params, pred, args, true, s1), lane_b(params, pred, args, true, s2);
    lane_a.prepare(make_block(0, 16)); lane_b.prepare(make_block(0, 16));
    TransformFinalAggregate final_(params);
    int lane = 0;
    for (int64_t lo = 0; lo < n; lo += BLOCK, ++lane) (lane % 2 ? lane_b : lane_a).transform(make_block(lo, std::min(n, lo + BLOCK)));
    for (auto* p : {&lane_a, &lane_b})
      for (auto& meta : p->on_finish(true)) final_.transform(std::move(meta));
    CHECK(lane_a.blocks_replayed() == 0 && lane_b.blocks_replayed() == 0);
    std::vector<DataBlock> res = final_.on_finish(true);
    CHECK(res.size() == 1);
    const DataBlock& r = res[0];
    struct G { int64_t q = 0, p = 0, d = 0; __int128 dp = 0, ch = 0; uint64_t c = 0; };
    std::map<std::pair<std::string, std::string>, G> exp;
    for (int64_t i = 0; i < n; ++i) if (ship[i] <= cutoff) {
      G& g = exp[{rf[i], ls[i]}];
      __int128 dp = (__int128)price[i] * (100 - disc[i]);
      g.q += qty[i]; g.p += price[i]; g.d += disc[i]; g.dp += dp; g.ch += dp * (100 + tax[i]); g.c += 1;
    }
    CHECK(r.num_rows == (int64_t)exp.size());
    auto sq = r.columns[0].to_vector<int64_t>(); auto sp = r.columns[1].to_vector<int64_t>();
    auto sdp = r.columns[2].to_vector<__int128>(); auto sch = r.columns[3].to_vector<__int128>();
    auto sd = r.columns[4].to_vector<int64_t>(); auto cnt = r.columns[5].to_vector<uint64_t>();
    auto krf = r.columns[6].to_short_strings(); auto kls = r.columns[7].to_short_strings();
    for (int64_t i = 0; i < r.num_rows; ++i) {
      auto it = exp.find({krf[(size_t)i], kls[(size_t)i]});
      CHECK(it != exp.end());
      if (it == exp.end()) continue;
      const G& g = it->second;
      CHECK(sq[i] == g.q && sp[i] == g.p && sd[i] == g.d && sdp[i] == g.dp && sch[i] == g.ch && cnt[i] == g.c);
    }
  }
  // ---- a block the fused kernel gives back: the operator replays it (and what was queued behind it) the slow way ----
  {
    const int64_t m = 30000, nb = 20;
    auto I64 = DataType::of(DBHIP_T_I64);
    AggregatorParams p2;
    p2.group_columns = {0}; p2.group_data_types = {I64};
    p2.aggregate_functions = {{"sum", 1, I64}, {"count", std::nullopt, DataType()}};
    Expr f2 = Expr::call("gte", {Expr::column_ref(1, I64, "x"), Expr::constant(Scalar::Int(DBHIP_T_I64, -500))});
    std::vector<std::optional<Expr>> a2 = {Expr::call("plus", {Expr::column_ref(1, I64, "x"), Expr::column_ref(1, I64, "x")}), std::nullopt};
    p2.aggregate_functions[0].arg_type = I64;
    TransformFusedPartialAggregate op(p2, f2, a2, true, s1);
    std::map<int64_t, std::pair<int64_t, uint64_t>> exp;
    for (int64_t b = 0; b < nb; ++b) {
      std::vector<int64_t> k(m), x(m);
      for (int64_t i = 0; i < m; ++i) { k[i] = (int64_t)(rng() % (b == 11 ? 60 : 3)); x[i] = (int64_t)(rng() % 2000) - 1000; }
      for (int64_t i = 0; i < m; ++i) if (x[i] >= -500) { exp[k[i]].first += 2 * x[i]; exp[k[i]].second += 1; }
      op.transform(DataBlock({Column::from_vector(I64, k), Column::from_vector(I64, x)}, m));
    }
    auto metas = op.on_finish(true);
    CHECK(op.blocks_replayed() == nb);   // one window: nothing of it committed
    DataBlock r = op.table().merge_result();
    CHECK(r.num_rows == (int64_t)exp.size());
    auto sx = r.columns[0].to_vector<int64_t>(); auto cx = r.columns[1].to_vector<uint64_t>(); auto kx = r.columns[2].to_vector<int64_t>();
    for (int64_t i = 0; i < r.num_rows; ++i) {
      auto it = exp.find(kx[(size_t)i]);
      CHECK(it != exp.end());
      if (it != exp.end()) CHECK(sx[(size_t)i] == it->second.first && cx[(size_t)i] == it->second.second);
    }
  }
  // ---- BlockAccumulator ----
  {
    BlockAccumulator acc(50000);
    std::vector<DataBlock> out;
    for (int64_t lo = 0; lo < 130000; lo += 20000) if (auto b = acc.add(make_block(lo, lo + 20000))) out.push_back(std::move(*b));
    if (auto b = acc.finish()) out.push_back(std::move(*b));
    CHECK(out.size() == 3 && out[0].num_rows == 60000 && out[1].num_rows == 60000 && out[2].num_rows == 20000);
    int64_t at = 0;
    for (const DataBlock& b : out) {
      auto q = b.columns[0].to_vector<int64_t>(); auto d = b.columns[6].to_vector<int32_t>(); auto f = b.columns[4].to_short_strings();
      for (int64_t i = 0; i < b.num_rows; ++i) CHECK(q[(size_t)i] == qty[at + i] && d[(size_t)i] == ship[at + i] && f[(size_t)i] == rf[at + i]);
      at += b.num_rows;
    }
  }
  check(dbhip_stream_destroy(s1)); check(dbhip_stream_destroy(s2));
}

static void test_join_and_sort() {
  const int64_t nb = 20000, np = 50000;
  std::mt19937_64 rng(5);
  std::vector<uint64_t> bk(nb), pk(np); std::vector<int64_t> bv(nb), pv(np);
  for (int64_t i = 0; i < nb; ++i) { bk[i] = rng() % 15000; bv[i] = i * 3; }
  for (int64_t i = 0; i < np; ++i) { pk[i] = rng() % 30000; pv[i] = -i; }
  auto U64 = DataType::of(DBHIP_T_U64); auto I64 = DataType::of(DBHIP_T_I64);
  InnerHashJoin join(0, 0, 8192);
  join.add_block(DataBlock({Column::from_vector(U64, bk), Column::from_vector(I64, bv)}, nb));
  join.add_block(std::nullopt);
  join.final_build();
  auto stream = join.probe_block(DataBlock({Column::from_vector(U64, pk), Column::from_vector(I64, pv)}, np));
  std::multiset<std::tuple<uint64_t, int64_t, int64_t>> got, exp;
  int64_t blocks = 0;
  while (auto b = stream->next()) {
    ++blocks;
    CHECK(b->num_rows <= 8192 && b->num_columns() == 4);
    auto k1 = b->columns[0].to_vector<uint64_t>(); auto v1 = b->columns[1].to_vector<int64_t>();
    auto k2 = b->columns[2].to_vector<uint64_t>(); auto v2 = b->columns[3].to_vector<int64_t>();
    for (size_t i = 0; i < k1.size(); ++i) { CHECK(k1[i] == k2[i]); got.insert({k1[i], v1[i], v2[i]}); }
  }
  std::multimap<uint64_t, int64_t> bm;
  for (int64_t i = 0; i < nb; ++i) bm.insert({bk[i], bv[i]});
  for (int64_t i = 0; i < np; ++i) { auto r = bm.equal_range(pk[i]); for (auto it = r.first; it != r.second; ++it) exp.insert({pk[i], pv[i], it->second}); }
  CHECK(got == exp && blocks > 1);
  // sort: two keys, desc + asc, limit
  std::vector<int32_t> k1(np); std::vector<double> k2(np);
  for (int64_t i = 0; i < np; ++i) { k1[i] = (int32_t)(rng() % 7); k2[i] = (double)(int64_t)(rng() % 100000) / 7.0; }
  DataBlock blk({Column::from_vector(DataType::of(DBHIP_T_I32), k1), Column::from_vector(DataType::of(DBHIP_T_F64), k2)}, np);
  DataBlock s = sort_block(blk, {{0, false, false}, {1, true, false}}, 1000);
  auto o1 = s.columns[0].to_vector<int32_t>(); auto o2 = s.columns[1].to_vector<double>();
  std::vector<std::pair<int32_t, double>> ref(np);
  for (int64_t i = 0; i < np; ++i) ref[i] = {-k1[i], k2[i]};
  std::sort(ref.begin(), ref.end());
  bool ok = s.num_rows == 1000;
  for (int64_t i = 0; i < 1000 && ok; ++i) ok &= o1[i] == -ref[i].first && o2[i] == ref[i].second;
  CHECK(ok);
}

static void test_left_joins() {
  const int64_t nb = 5000, np = 20000;
  std::mt19937_64 rng(9);
  std::vector<uint64_t> bk(nb), pk(np); std::vector<int64_t> bv(nb), pv(np);
  for (int64_t i = 0; i < nb; ++i) { bk[i] = rng() % 4000; bv[i] = i * 7 + 1; }
  for (int64_t i = 0; i < np; ++i) { pk[i] = rng() % 8000; pv[i] = i; }
  auto U64 = DataType::of(DBHIP_T_U64); auto I64 = DataType::of(DBHIP_T_I64);
  std::multimap<uint64_t, int64_t> bm;
  for (int64_t i = 0; i < nb; ++i) bm.insert({bk[i], bv[i]});
  for (LeftJoinKind kind : {LeftJoinKind::Outer, LeftJoinKind::Semi, LeftJoinKind::Anti}) {
    LeftHashJoin join(kind, 0, 0);
    join.add_block(DataBlock({Column::from_vector(U64, bk), Column::from_vector(I64, bv)}, nb));
    join.final_build();
    auto stream = join.probe_block(DataBlock({Column::from_vector(U64, pk), Column::from_vector(I64, pv)}, np));
    auto b = stream->next();
    CHECK(b.has_value() && !stream->next().has_value());
    auto k1 = b->columns[0].to_vector<uint64_t>(); auto v1 = b->columns[1].to_vector<int64_t>();
    if (kind != LeftJoinKind::Outer) {
      CHECK(b->num_columns() == 2);
      std::vector<int64_t> exp;
      for (int64_t i = 0; i < np; ++i) if ((bm.count(pk[i]) > 0) == (kind == LeftJoinKind::Semi)) exp.push_back(pv[i]);
      CHECK(v1 == exp);
      continue;
    }
    CHECK(b->num_columns() == 4 && b->columns[3].type.nullable && b->columns[3].validity);
    auto v2 = b->columns[3].to_vector<int64_t>();
    std::vector<uint8_t> vb((size_t)(b->num_rows + 7) / 8);
    b->columns[3].validity->download(vb.data(), vb.size());
    std::multiset<std::tuple<uint64_t, int64_t, int64_t>> got, exp;   // build value -1 = NULL
    for (size_t i = 0; i < k1.size(); ++i) got.insert({k1[i], v1[i], ((vb[i >> 3] >> (i & 7)) & 1) ? v2[i] : -1});
    for (int64_t i = 0; i < np; ++i) {
      auto r = bm.equal_range(pk[i]);
      if (r.first == r.second) exp.insert({pk[i], pv[i], -1});
      for (auto it = r.first; it != r.second; ++it) exp.insert({pk[i], pv[i], it->second});
    }
    CHECK(got == exp);
  }
}

static void test_join_conjuncts() {
  // the CONJUNCT = true streams: key equality AND `probe value < build value` (a nullable build value: NULL drops the pair);
  // inner / left outer / left semi / left anti / right outer / right anti against a multimap statement
  const int64_t nb = 4000, np = 9000;
  std::mt19937_64 rng(23);
  std::vector<uint64_t> bk(nb), pk(np); std::vector<int64_t> bv(nb), pv(np); std::vector<bool> bvalid(nb);
  for (int64_t i = 0; i < nb; ++i) { bk[i] = rng() % 3000; bv[i] = (int64_t)(rng() % 1000); bvalid[i] = rng() % 5 != 0; }
  for (int64_t i = 0; i < np; ++i) { pk[i] = rng() % 4500; pv[i] = (int64_t)(rng() % 1000); }
  auto U64 = DataType::of(DBHIP_T_U64); auto I64 = DataType::of(DBHIP_T_I64); auto U32 = DataType::of(DBHIP_T_U32);
  std::vector<uint32_t> bid(nb), pid(np);
  for (int64_t i = 0; i < nb; ++i) bid[i] = (uint32_t)i;
  for (int64_t i = 0; i < np; ++i) pid[i] = (uint32_t)i;
  Column bval = Column::from_vector(I64, bv);
  bval.validity = Column::from_bools(bvalid).data; bval.type.nullable = true;
  DataBlock build({Column::from_vector(U64, bk), bval, Column::from_vector(U32, bid)}, nb);
  auto probe = [&]() { return DataBlock({Column::from_vector(U64, pk), Column::from_vector(I64, pv), Column::from_vector(U32, pid)}, np); };
  // joined block = probe columns (0..2) ++ build columns (3..5)
  JoinConjunct conj = [](const DataBlock& j) {
    Evaluator ev(j);
    return ev.run(Expr::call("lt", {Expr::column_ref(1, j.columns[1].type, "p"), Expr::column_ref(4, j.columns[4].type, "b")})).column;
  };
  std::multimap<uint64_t, int64_t> bm;
  for (int64_t i = 0; i < nb; ++i) bm.insert({bk[i], i});
  std::set<std::pair<int64_t, int64_t>> pairs;
  std::vector<bool> pm(np, false), bmatched(nb, false);
  for (int64_t i = 0; i < np; ++i) {
    auto r = bm.equal_range(pk[i]);
    for (auto it = r.first; it != r.second; ++it)
      if (bvalid[it->second] && pv[i] < bv[it->second]) { pairs.insert({i, it->second}); pm[i] = true; bmatched[it->second] = true; }
  }
  CHECK(!pairs.empty());
  auto ids = [](const Column& c, int64_t n) {   // row ids, -1 where NULL
    auto v = c.to_vector<uint32_t>();
    std::vector<uint8_t> bits((size_t)(n + 7) / 8 + 8, 0xFF);
    if (c.validity) c.validity->download(bits.data(), (size_t)(n + 7) / 8);
    std::vector<int64_t> o((size_t)n);
    for (int64_t i = 0; i < n; ++i) o[i] = ((bits[i >> 3] >> (i & 7)) & 1) ? (int64_t)v[i] : -1;
    return o;
  };
  {  // inner
    InnerHashJoin j(0, 0);
    j.set_conjunct(conj);
    j.add_block(build); j.final_build();
    auto st = j.probe_block(probe());
    std::set<std::pair<int64_t, int64_t>> got; int64_t rows = 0;
    while (auto b = st->next()) {
      auto p = ids(b->columns[2], b->num_rows), q = ids(b->columns[5], b->num_rows);
      for (int64_t i = 0; i < b->num_rows; ++i) got.insert({p[i], q[i]});
      rows += b->num_rows;
    }
    CHECK(got == pairs && rows == (int64_t)pairs.size());
  }
  for (LeftJoinKind kind : {LeftJoinKind::Outer, LeftJoinKind::Semi, LeftJoinKind::Anti}) {
    LeftHashJoin j(kind, 0, 0);
    j.set_conjunct(conj);
    j.add_block(build); j.final_build();
    auto st = j.probe_block(probe());
    auto b = st->next();
    CHECK(b.has_value());
    auto p = ids(b->columns[2], b->num_rows);
    if (kind == LeftJoinKind::Outer) {
      auto q = ids(b->columns[5], b->num_rows);
      std::multiset<std::pair<int64_t, int64_t>> got, exp(pairs.begin(), pairs.end());
      for (int64_t i = 0; i < b->num_rows; ++i) got.insert({p[i], q[i]});
      for (int64_t i = 0; i < np; ++i) if (!pm[i]) exp.insert({i, -1});
      CHECK(got == exp);
    } else {
      std::vector<int64_t> exp;
      for (int64_t i = 0; i < np; ++i) if (pm[i] == (kind == LeftJoinKind::Semi)) exp.push_back(i);
      CHECK(p == exp);
    }
  }
  for (RightJoinKind kind : {RightJoinKind::Outer, RightJoinKind::Anti}) {
    RightHashJoin j(kind, 0, 0);
    j.set_conjunct(conj);
    j.add_block(build); j.final_build();
    std::multiset<std::pair<int64_t, int64_t>> got, exp;
    auto st = j.probe_block(probe());
    while (auto b = st->next()) {
      auto p = ids(b->columns[2], b->num_rows), q = ids(b->columns[5], b->num_rows);
      for (int64_t i = 0; i < b->num_rows; ++i) got.insert({p[i], q[i]});
    }
    auto tail = j.final_probe();
    CHECK(tail.has_value());
    if (kind == RightJoinKind::Anti) {
      auto q = ids(tail->columns[2], tail->num_rows);
      std::vector<int64_t> want;
      for (int64_t i = 0; i < nb; ++i) if (!bmatched[i]) want.push_back(i);
      CHECK(got.empty() && q == want);
      continue;
    }
    auto q = ids(tail->columns[5], tail->num_rows), p = ids(tail->columns[2], tail->num_rows);
    for (int64_t i = 0; i < tail->num_rows; ++i) got.insert({p[i], q[i]});
    exp.insert(pairs.begin(), pairs.end());
    for (int64_t i = 0; i < nb; ++i) if (!bmatched[i]) exp.insert({-1, i});
    CHECK(got == exp);
  }
}

static void test_flight_scatters() {
  // HashFlightScatter: every row lands in the block of siphash64(key) % n (NULL keys in block `default`), nothing is lost;
  // SortBoundScatter: block i holds the rows after bound i - 1 up to and including bound i
  const int64_t n = 20000; const size_t parts = 5;
  std::mt19937_64 rng(31);
  std::vector<int64_t> key(n), pay(n); std::vector<bool> valid(n);
  for (int64_t i = 0; i < n; ++i) { key[i] = (int64_t)(rng() % 3000) - 1500; pay[i] = i; valid[i] = rng() % 7 != 0; }
  auto I64 = DataType::of(DBHIP_T_I64);
  Column kc = Column::from_vector(I64, key);
  kc.validity = Column::from_bools(valid).data; kc.type.nullable = true;
  DataBlock block({kc, Column::from_vector(I64, pay)}, n);
  // the hashes, through the scalar function
  Buf hb = make_buf((size_t)n * 8);
  dbhip_col kcol = kc.c();
  check(dbhip_siphash64(&kcol, n, (uint64_t*)hb->ptr(), nullptr));
  std::vector<uint64_t> h((size_t)n);
  hb->download(h.data(), (size_t)n * 8);
  HashFlightScatter sc({0}, parts, 2);
  CHECK(std::string(sc.name()) == "OneHashKey");
  auto blocks = sc.execute(block);
  CHECK(blocks.size() == parts);
  int64_t seen = 0;
  for (size_t d = 0; d < parts; ++d) {
    auto p = blocks[d].columns[1].to_vector<int64_t>();
    int64_t prev = -1;
    for (int64_t r : p) {
      CHECK((valid[r] ? h[r] % parts : 2) == d);
      CHECK(r > prev);    // DataBlock::scatter keeps the row order inside a destination
      prev = r;
    }
    seen += (int64_t)p.size();
  }
  CHECK(seen == n);
  // bounds -100 and 700 (ascending, NULLs last): three ranges
  SortBoundScatter sb({SortColumnDescription{0, true, false}}, DataBlock({Column::from_vector(I64, std::vector<int64_t>{-100, 700})}, 2));
  auto ranges = sb.execute(block);
  CHECK(ranges.size() == 3);
  seen = 0;
  for (size_t d = 0; d < 3; ++d) {
    for (int64_t r : ranges[d].columns[1].to_vector<int64_t>()) {
      const size_t want = !valid[r] ? 2 : key[r] <= -100 ? 0 : key[r] <= 700 ? 1 : 2;
      CHECK(want == d);
      ++seen;
    }
  }
  CHECK(seen == n);
}

// The ABI-owned exchange between TWO ranks without Python or torch: two host threads, one loopback communicator each (an in-process
// world, dbhip_comm_create_loopback), every rank shuffles its shard by siphash64(key) % 2 and both exchange in one grouped all-to-all
// (dbhip_exchange_begin / _finish); then the shard top-k merge (dbhip_vec_topk_allgather). Checked against host loops.
static void test_two_rank_exchange() {
  const int world = 2;
  const int64_t n[2] = {30000, 17001};
  std::vector<int64_t> key[2], pay[2];
  std::vector<uint32_t> dest[2];
  std::vector<int64_t> got_key[2], got_pay[2];
  std::vector<int64_t> shuf_key[2], shuf_pay[2], sort_key[2], sort_pay[2];   // the same block through the two plan calls
  std::vector<int64_t> starts[2];
  std::vector<std::string> tag[2], got_tag[2];     // a String column whose values are mostly longer than 12 bytes (they travel packed)
  std::mt19937_64 rng(77);
  for (int r = 0; r < world; ++r) {
    key[r].resize(n[r]); pay[r].resize(n[r]);
    for (int64_t i = 0; i < n[r]; ++i) { key[r][i] = (int64_t)(rng() % 4096); pay[r][i] = (int64_t)r * 1000000 + i; }
    tag[r].resize(n[r]);
    for (int64_t i = 0; i < n[r]; ++i)
      tag[r][i] = i % 7 == 0 ? std::string("t") + std::to_string(i % 90) : "rank-" + std::to_string(r) + "-row-" + std::to_string(i) + std::string((size_t)(i % 40), 'y');
  }
  const int nq = 5, k = 4;
  std::vector<uint32_t> tid[2], gid[2];
  std::vector<float> tdist[2], gdist[2];
  for (int r = 0; r < world; ++r) {
    tid[r].resize(nq * k); tdist[r].resize(nq * k); gid[r].resize(nq * k); gdist[r].resize(nq * k);
    for (int q = 0; q < nq; ++q)
      for (int j = 0; j < k; ++j) { tid[r][q * k + j] = (uint32_t)(q * 100 + j * 7 + r); tdist[r][q * k + j] = 0.1f * j + 0.03f * r + 0.001f * q; }
  }
  std::string err[2];
  auto rank_fn = [&](int r) {
    try {
      dbhip_comm* c = nullptr;
      check(dbhip_comm_create_loopback(4242, r, world, &c));
      auto I64 = DataType::of(DBHIP_T_I64);
      Column kc = Column::from_vector(I64, key[r]), pc = Column::from_vector(I64, pay[r]);
      Buf db = make_buf((size_t)n[r] * 4 + 64), cb = make_buf((size_t)world * 8);
      dbhip_col kcol = kc.c();
      check(dbhip_scatter_indices(&kcol, 1, n[r], (uint32_t)world, 0, (uint32_t*)db->ptr(), (uint64_t*)cb->ptr(), nullptr));
      dest[r].resize(n[r]);
      db->download(dest[r].data(), (size_t)n[r] * 4);
      dbhip_col cols[2] = {kc.c(), pc.c()};
      int64_t rows = 0;
      dbhip_exchange* x = nullptr;
      check(dbhip_exchange_begin(c, cols, 2, (const uint32_t*)db->ptr(), n[r], &rows, &x, nullptr));
      Buf ok = make_buf((size_t)rows * 8 + 64), op = make_buf((size_t)rows * 8 + 64);
      void* outs[2] = {ok->ptr(), op->ptr()};
      uint8_t* vouts[2] = {nullptr, nullptr};
      starts[r].resize(world + 1);
      check(dbhip_exchange_finish(x, outs, vouts, starts[r].data(), nullptr));
      check(dbhip_exchange_destroy(x));
      got_key[r].resize(rows); got_pay[r].resize(rows);
      ok->download(got_key[r].data(), (size_t)rows * 8);
      op->download(got_pay[r].data(), (size_t)rows * 8);
      {   // the same rows with a String column beside the key: long values are packed per destination and re-based at the receiver
        Column tc = Column::from_strings(tag[r]);
        dbhip_col scols[2] = {kc.c(), tc.c()};
        int64_t srows = 0;
        dbhip_exchange* sx = nullptr;
        check(dbhip_exchange_begin(c, scols, 2, (const uint32_t*)db->ptr(), n[r], &srows, &sx, nullptr));
        int64_t sbytes[2] = {0, 0};
        check(dbhip_exchange_string_bytes(sx, sbytes));
        Buf sk = make_buf((size_t)srows * 8 + 64), sv = make_buf((size_t)srows * 16 + 64), sb = make_buf((size_t)sbytes[1] + 64);
        void* souts[2] = {sk->ptr(), sv->ptr()};
        uint8_t* svalid[2] = {nullptr, nullptr};
        uint8_t* sbufs[2] = {nullptr, (uint8_t*)sb->ptr()};
        check(dbhip_exchange_finish_strings(sx, souts, svalid, sbufs, nullptr, nullptr));
        check(dbhip_exchange_destroy(sx));
        got_tag[r] = Column::strings_to_host(sv->ptr(), srows, sb->ptr(), sbytes[1]);
      }
      // the hash shuffle and the range partition of the distributed sort as single plan calls (no index computed by the caller)
      for (int plan = 0; plan < 2; ++plan) {
        int64_t prow = 0;
        dbhip_exchange* px = nullptr;
        if (plan == 0) {
          check(dbhip_shuffle_exchange_begin(c, &kcol, 1, cols, 2, n[r], &prow, &px, nullptr));
        } else {
          const std::vector<int64_t> bound = {2047};    // one bound: keys <= 2047 -> rank 0, the rest -> rank 1
          Column bc = Column::from_vector(I64, bound);
          dbhip_col bcol = bc.c();
          const uint8_t desc[1] = {0}, nf[1] = {0};
          check(dbhip_sort_exchange_begin(c, &kcol, &bcol, desc, nf, 1, 1, cols, 2, n[r], &prow, &px, nullptr));
        }
        Buf pk = make_buf((size_t)prow * 8 + 64), pp = make_buf((size_t)prow * 8 + 64);
        void* pouts[2] = {pk->ptr(), pp->ptr()};
        uint8_t* pv[2] = {nullptr, nullptr};
        check(dbhip_exchange_finish(px, pouts, pv, nullptr, nullptr));
        check(dbhip_exchange_destroy(px));
        std::vector<int64_t>& dk = plan == 0 ? shuf_key[r] : sort_key[r];
        std::vector<int64_t>& dp = plan == 0 ? shuf_pay[r] : sort_pay[r];
        dk.resize(prow); dp.resize(prow);
        pk->download(dk.data(), (size_t)prow * 8);
        pp->download(dp.data(), (size_t)prow * 8);
      }
      Buf ib = make_buf(nq * k * 4), dbf = make_buf(nq * k * 4), oi = make_buf(nq * k * 4), od = make_buf(nq * k * 4);
      ib->upload(tid[r].data(), nq * k * 4); dbf->upload(tdist[r].data(), nq * k * 4);
      check(dbhip_vec_topk_allgather(c, (const uint32_t*)ib->ptr(), (const float*)dbf->ptr(), nq, k, (uint64_t)r * 5000000ULL, (uint32_t*)oi->ptr(),
                                     (float*)od->ptr(), nullptr));
      oi->download(gid[r].data(), nq * k * 4); od->download(gdist[r].data(), nq * k * 4);
      check(dbhip_comm_destroy(c));
    } catch (const std::exception& e) { err[r] = e.what(); }
  };
  std::thread t0(rank_fn, 0), t1(rank_fn, 1);
  t0.join(); t1.join();
  CHECK(err[0].empty() && err[1].empty());
  if (!err[0].empty() || !err[1].empty()) { printf("  exchange threads: %s | %s\n", err[0].c_str(), err[1].c_str()); return; }
  for (int r = 0; r < world; ++r) {
    std::vector<int64_t> ek, ep;
    for (int s = 0; s < world; ++s)
      for (int64_t i = 0; i < n[s]; ++i)
        if ((int)dest[s][i] == r) { ek.push_back(key[s][i]); ep.push_back(pay[s][i]); }
    CHECK(got_key[r] == ek);
    CHECK(got_pay[r] == ep);
    {
      std::vector<std::string> et;
      for (int s = 0; s < world; ++s)
        for (int64_t i = 0; i < n[s]; ++i)
          if ((int)dest[s][i] == r) et.push_back(tag[s][i]);
      CHECK(got_tag[r] == et);
    }
    CHECK(starts[r][world] == (int64_t)ek.size());
    CHECK(shuf_key[r] == ek && shuf_pay[r] == ep);      // dbhip_shuffle_exchange_begin = scatter_indices + exchange
    std::vector<int64_t> sk, sp;                        // dbhip_sort_exchange_begin: the rows of range r, by source rank, in order
    for (int s2 = 0; s2 < world; ++s2)
      for (int64_t i = 0; i < n[s2]; ++i)
        if ((key[s2][i] <= 2047 ? 0 : 1) == r) { sk.push_back(key[s2][i]); sp.push_back(pay[s2][i]); }
    CHECK(sort_key[r] == sk && sort_pay[r] == sp);
    for (int q = 0; q < nq; ++q) {   // the k smallest of both shards' candidates, ids made global
      std::vector<std::pair<float, uint32_t>> all;
      for (int s = 0; s < world; ++s)
        for (int j = 0; j < k; ++j) all.push_back({tdist[s][q * k + j], tid[s][q * k + j] + (uint32_t)(s * 5000000)});
      std::sort(all.begin(), all.end());
      for (int j = 0; j < k; ++j) { CHECK(gid[r][q * k + j] == all[j].second); CHECK(gdist[r][q * k + j] == all[j].first); }
    }
  }
  CHECK((int64_t)(got_key[0].size() + got_key[1].size()) == n[0] + n[1]);
}

static void test_hnsw_sequential_build_and_open() {
  // the deterministic build gives the same graph twice; store() -> open() searches identically without the original vectors
  const int dim = 12; const int64_t n = 1500;
  std::mt19937 rng(8);
  std::normal_distribution<float> nd(0.f, 1.f);
  std::vector<float> base((size_t)n * dim);
  for (auto& x : base) x = nd(rng);
  std::vector<int32_t> levels((size_t)n);
  for (auto& l : levels) { const double u = ((double)(rng() % 1000000) + 0.5) / 1000000.0; l = (int32_t)std::llround(-std::log(u) / std::log(8.0)); }
  Column col = Column::from_vector(DataType::Vector(dim), base);
  HNSWIndex a = HNSWIndex::build_sequential(8, 32, col, DBHIP_VEC_L2, levels), b = HNSWIndex::build_sequential(8, 32, col, DBHIP_VEC_L2, levels);
  HNSWIndex::Stored sa = a.store(), sb = b.store();
  CHECK(sa.levels == levels && sa.links == sb.links && sa.nlinks == sb.nlinks && sa.entry_point == sb.entry_point && sa.encoded == sb.encoded);
  HNSWIndex o = HNSWIndex::open(DBHIP_VEC_L2, dim, 8, sa);
  std::vector<float> q((size_t)64 * dim);
  for (auto& x : q) x = nd(rng);
  Column qc = Column::from_vector(DataType::Vector(dim), q);
  auto r1 = a.search(10, qc), r2 = o.search(10, qc);
  CHECK(r1.first == r2.first);
  for (size_t i = 0; i < r1.second.size(); ++i) CHECK(r1.second[i] == r2.second[i] || (r1.second[i] != r1.second[i] && r2.second[i] != r2.second[i]));
}

static void test_right_joins() {
  // right / right-semi / right-anti / full over TWO probe blocks (the scan map lives across blocks), against std::multimap
  const int64_t nb = 5000, np = 12000;
  std::mt19937_64 rng(19);
  std::vector<uint64_t> bk(nb), pk(np); std::vector<int64_t> bv(nb), pv(np);
  for (int64_t i = 0; i < nb; ++i) { bk[i] = rng() % 6000; bv[i] = i * 5 + 2; }
  for (int64_t i = 0; i < np; ++i) { pk[i] = rng() % 9000; pv[i] = i + 1; }
  auto U64 = DataType::of(DBHIP_T_U64); auto I64 = DataType::of(DBHIP_T_I64);
  std::multimap<uint64_t, int64_t> pm;
  for (int64_t i = 0; i < np; ++i) pm.insert({pk[i], pv[i]});
  std::multimap<uint64_t, int64_t> bm;
  for (int64_t i = 0; i < nb; ++i) bm.insert({bk[i], bv[i]});
  auto valid_at = [](const Column& c, size_t i) {
    if (!c.validity) return true;
    std::vector<uint8_t> vb((size_t)(c.len + 7) / 8 + 8);
    c.validity->download(vb.data(), (size_t)(c.len + 7) / 8);
    return (bool)((vb[i >> 3] >> (i & 7)) & 1);
  };
  for (RightJoinKind kind : {RightJoinKind::Outer, RightJoinKind::Semi, RightJoinKind::Anti, RightJoinKind::Full}) {
    RightHashJoin join(kind, 0, 0);
    join.add_block(DataBlock({Column::from_vector(U64, bk), Column::from_vector(I64, bv)}, nb));
    join.final_build();
    std::multiset<std::tuple<int64_t, int64_t>> got, exp;   // (probe value or -1 = NULL, build value or -1 = NULL)
    auto collect = [&](const DataBlock& b) {
      if (b.num_rows == 0) return;
      CHECK(b.num_columns() == 4);
      auto v1 = b.columns[1].to_vector<int64_t>(); auto v2 = b.columns[3].to_vector<int64_t>();
      std::vector<uint8_t> p1((size_t)(b.num_rows + 7) / 8 + 8, 0xFF), p2((size_t)(b.num_rows + 7) / 8 + 8, 0xFF);
      if (b.columns[1].validity) b.columns[1].validity->download(p1.data(), (size_t)(b.num_rows + 7) / 8);
      if (b.columns[3].validity) b.columns[3].validity->download(p2.data(), (size_t)(b.num_rows + 7) / 8);
      for (size_t i = 0; i < (size_t)b.num_rows; ++i)
        got.insert({((p1[i >> 3] >> (i & 7)) & 1) ? v1[i] : -1, ((p2[i >> 3] >> (i & 7)) & 1) ? v2[i] : -1});
    };
    std::multiset<int64_t> got1;   // semi / anti: build values
    for (int half = 0; half < 2; ++half) {
      const int64_t lo = half * (np / 2), hi = half ? np : np / 2;
      DataBlock pb({Column::from_vector(U64, std::vector<uint64_t>(pk.begin() + lo, pk.begin() + hi)),
                    Column::from_vector(I64, std::vector<int64_t>(pv.begin() + lo, pv.begin() + hi))}, hi - lo);
      auto stream = join.probe_block(std::move(pb));
      while (auto b = stream->next()) collect(*b);
    }
    auto tail = join.final_probe();
    if (kind == RightJoinKind::Semi || kind == RightJoinKind::Anti) {
      CHECK(got.empty() && tail.has_value() && tail->num_columns() == 2);
      for (int64_t v : tail->columns[1].to_vector<int64_t>()) got1.insert(v);
      std::multiset<int64_t> exp1;
      for (int64_t i = 0; i < nb; ++i) if ((pm.count(bk[i]) > 0) == (kind == RightJoinKind::Semi)) exp1.insert(bv[i]);
      CHECK(got1 == exp1);
      continue;
    }
    if (tail) collect(*tail);
    for (int64_t i = 0; i < np; ++i) {
      auto r = bm.equal_range(pk[i]);
      if (r.first == r.second && kind == RightJoinKind::Full) exp.insert({pv[i], -1});
      for (auto it = r.first; it != r.second; ++it) exp.insert({pv[i], it->second});
    }
    for (int64_t i = 0; i < nb; ++i) if (pm.count(bk[i]) == 0) exp.insert({-1, bv[i]});
    CHECK(got == exp);
    (void)valid_at;
  }
}

static void test_kmeans() {
  // TransformVectorCluster's KMeans: deterministic (fixed seed), every row assigned to its nearest centroid, k = ceil(n / rows_per_cluster)
  const int dim = 8; const int64_t n = 4000;
  std::mt19937 rng(6);
  std::normal_distribution<float> nd(0.f, 0.3f);
  std::vector<float> data((size_t)n * dim);
  for (int64_t i = 0; i < n; ++i) for (int d = 0; d < dim; ++d) data[(size_t)i * dim + d] = (float)((i % 5) * 4) + nd(rng);   // 5 well separated blobs
  Column col = Column::from_vector(DataType::Vector(dim), data);
  KMeansResult a = kmeans(1, col, 800, false), b = kmeans(1, col, 800, false);
  CHECK(a.k == 5 && a.iterations >= 1 && a.assignments == b.assignments && a.distances == b.distances);
  std::set<std::pair<int, uint32_t>> pairs;   // every blob in exactly one cluster
  for (int64_t i = 0; i < n; ++i) pairs.insert({(int)(i % 5), a.assignments[(size_t)i]});
  CHECK(pairs.size() == 5);
  for (float d : a.distances) CHECK(d >= 0.f && d < 3.f);
}

static void test_hnsw_index() {
  // HNSWIndex::build + search (m = 10, ef_construct = 40, ef = 4 k): every vector finds itself, distances ascend
  const int dim = 16; const int64_t n = 3000;
  std::mt19937 rng(4);
  std::normal_distribution<float> nd(0.f, 1.f);
  std::vector<float> base((size_t)n * dim);
  for (auto& x : base) x = nd(rng);
  Column col = Column::from_vector(DataType::Vector(dim), base);
  col.len = n;
  HNSWIndex ix = HNSWIndex::build(10, 40, col, DBHIP_VEC_L2);
  std::vector<float> q(base.begin(), base.begin() + 64 * dim);
  Column qc = Column::from_vector(DataType::Vector(dim), q);
  qc.len = 64;
  auto r = ix.search(5, qc);
  int self = 0;
  bool ascending = true;
  for (int i = 0; i < 64; ++i) {
    self += r.first[(size_t)i * 5] == (uint32_t)i;
    for (int j = 1; j < 5; ++j) ascending &= r.second[(size_t)i * 5 + j] >= r.second[(size_t)i * 5 + j - 1];
  }
  CHECK(self >= 60 && ascending);
}

static void test_vector_function() {
  const int dim = 8; const int64_t n = 16;
  std::mt19937 rng(8);
  std::vector<float> base(n * dim), q(dim);
  for (auto& x : base) x = (float)(rng() % 1000) / 1000.f;
  for (auto& x : q) x = (float)(rng() % 1000) / 1000.f;
  DataBlock block({Column::from_vector(DataType::Vector(dim), base), Column::from_vector(DataType::Vector(dim), q)}, n);
  Expr e = Expr::call("cosine_distance", {Expr::column_ref(0, DataType::Vector(dim), "embedding"), Expr::column_ref(1, DataType::Vector(dim), "q")});
  // the query column holds ONE vector (a constant in the plan)
  DataBlock b2 = block; b2.columns[1].len = 1; b2.columns[1].is_const = true;
  Evaluator ev(b2);
  Value v = ev.run(e);
  auto got = v.column.to_vector<float>();
  for (int64_t i = 0; i < n; ++i) {
    double ab = 0, aa = 0, bb = 0;
    for (int k = 0; k < dim; ++k) { ab += (double)base[i * dim + k] * q[k]; aa += (double)base[i * dim + k] * base[i * dim + k]; bb += (double)q[k] * q[k]; }
    double exp = 1.0 - ab / (std::sqrt(aa) * std::sqrt(bb));
    CHECK(std::fabs(got[i] - exp) <= 1e-5 * std::max(1.0, std::fabs(exp)) + 1e-6);
  }
}


// a hand-assembled column chunk: one DATA_PAGE (v1), PLAIN, optional INT64, values {7, NULL, -3, 1<<40}
static void test_parquet_chunk() {
  std::vector<uint8_t> ch = {
      0x15, 0x00,              // PageHeader.type = DATA_PAGE
      0x15, 0x3C,              // uncompressed_page_size = 30 (zigzag 60): 4 + 2 level bytes + 3 x 8 value bytes
      0x15, 0x3C,              // compressed_page_size = 30
      0x2C,                    // field 5: data_page_header {
      0x15, 0x08,              //   num_values = 4
      0x15, 0x00,              //   encoding = PLAIN
      0x15, 0x06,              //   definition_level_encoding = RLE
      0x15, 0x06,              //   repetition_level_encoding = RLE
      0x00,                    // }
      0x00,                    // end of PageHeader
      // definition levels: u32 length = 2, then one bit-packed run of 1 group (header 0x03), bits 1,0,1,1 -> 0x0D
      0x02, 0x00, 0x00, 0x00, 0x03, 0x0D,
  };
  const int64_t vals[3] = {7, -3, (int64_t)1 << 40};
  const uint8_t* vb = (const uint8_t*)vals;
  ch.insert(ch.end(), vb, vb + 24);
  ParquetLeaf leaf{2 /*INT64*/, 0, 1, 0, 0};
  auto col = column_chunk_to_column(ch.data(), ch.size(), leaf, DataType::of(DBHIP_T_I64));
  CHECK(col.has_value());
  if (col) {
    CHECK(col->len == 4);
    auto v = col->to_vector<int64_t>();
    auto ok = col->validity_to_host();
    CHECK(v[0] == 7 && v[1] == 0 && v[2] == -3 && v[3] == ((int64_t)1 << 40));
    CHECK(ok[0] && !ok[1] && ok[2] && ok[3]);
  }
  leaf.codec = 2;  // GZIP: declined, the caller keeps the CPU reader
  CHECK(!column_chunk_to_column(ch.data(), ch.size(), leaf, DataType::of(DBHIP_T_I64)).has_value());
}

// the device-mode scan of a block (two leaves in one dbhip_pq_chunks_decode_device call) and a List<Int64> leaf; pages written by hand
static void test_parquet_device_mode() {
  std::vector<uint8_t> ch = {
      0x15, 0x00, 0x15, 0x3C, 0x15, 0x3C, 0x2C, 0x15, 0x08, 0x15, 0x00, 0x15, 0x06, 0x15, 0x06, 0x00, 0x00,   // DATA_PAGE, 30 bytes, 4 values, PLAIN, RLE levels
      0x02, 0x00, 0x00, 0x00, 0x03, 0x0D,                                                                 // definition levels 1,0,1,1
  };
  const int64_t vals[3] = {7, -3, (int64_t)1 << 40};
  ch.insert(ch.end(), (const uint8_t*)vals, (const uint8_t*)vals + 24);
  std::vector<uint8_t> req = {
      0x15, 0x00, 0x15, 0x20, 0x15, 0x20, 0x2C, 0x15, 0x04, 0x15, 0x00, 0x15, 0x06, 0x15, 0x06, 0x00, 0x00,   // a required INT64 column: 16 bytes, 2 values
  };
  const int64_t two[2] = {11, 12};
  req.insert(req.end(), (const uint8_t*)two, (const uint8_t*)two + 16);
  auto cols = column_chunks_to_columns({{ch.data(), ch.size()}, {req.data(), req.size()}}, {ParquetLeaf{2, 0, 1, 0, 0}, ParquetLeaf{2, 0, 0, 0, 0}},
                                       {DataType::of(DBHIP_T_I64), DataType::of(DBHIP_T_I64)});
  CHECK(cols.size() == 2 && cols[0].has_value() && cols[1].has_value());
  if (cols[0] && cols[1]) {
    auto v = cols[0]->to_vector<int64_t>();
    auto ok = cols[0]->validity_to_host();
    CHECK(cols[0]->len == 4 && v[0] == 7 && v[1] == 0 && v[2] == -3 && v[3] == ((int64_t)1 << 40) && ok[0] && !ok[1] && ok[2] && ok[3]);
    auto w = cols[1]->to_vector<int64_t>();
    CHECK(cols[1]->len == 2 && w[0] == 11 && w[1] == 12);
  }
  // List<Int64>, list and elements nullable (max_def 3): rows [5, NULL], NULL, [], [6] -> entries (rep, def): (0,3) (1,2) (0,0) (0,1) (0,3)
  std::vector<uint8_t> li = {
      0x15, 0x00, 0x15, 0x3A, 0x15, 0x3A, 0x2C, 0x15, 0x0A, 0x15, 0x00, 0x15, 0x06, 0x15, 0x06, 0x00, 0x00,   // DATA_PAGE, 29 bytes, 5 entries
      0x02, 0x00, 0x00, 0x00, 0x03, 0x02,          // repetition levels, width 1: one bit-packed group, bits 0,1,0,0,0 -> 0x02
      0x03, 0x00, 0x00, 0x00, 0x03, 0x4B, 0x03,    // definition levels, width 2: one group of 8: 3,2,0,1,3,0,0,0 -> bytes 0x4B 0x03
  };
  const int64_t el[2] = {5, 6};
  li.insert(li.end(), (const uint8_t*)el, (const uint8_t*)el + 16);
  auto L = list_chunk_to_column(li.data(), li.size(), ParquetLeaf{2, 0, 3, 1, 0}, true, true, DataType::of(DBHIP_T_I64));
  CHECK(L.has_value());
  if (L) {
    CHECK(L->offsets == (std::vector<uint64_t>{0, 2, 2, 2, 3}));
    CHECK(L->list_valid == (std::vector<uint8_t>{1, 0, 1, 1}));
    auto v = L->elements.to_vector<int64_t>();
    auto ok = L->elements.validity_to_host();
    CHECK(L->elements.len == 3 && v[0] == 5 && v[1] == 0 && v[2] == 6 && ok[0] && !ok[1] && ok[2]);
  }
}

int main() {
  try {
    init(0);
    test_sum_a_plus_b_mul_c();
    test_row_errors();
    test_filter();
    test_or_filters();
    test_selector();
    test_q1_plan();
    test_fused_partial_aggregate();
    test_join_and_sort();
    test_left_joins();
    test_hnsw_sequential_build_and_open();
    test_join_conjuncts();
    test_flight_scatters();
    test_two_rank_exchange();
    test_right_joins();
    test_kmeans();
    test_hnsw_index();
    test_vector_function();
    test_parquet_chunk();
    test_parquet_device_mode();
  } catch (const std::exception& e) {
    printf("EXCEPTION: %s\n", e.what());
    return 2;
  }
  printf(g_fail ? "host_selftest: %d check(s) FAILED\n" : "host_selftest: all checks passed\n", g_fail);
  return g_fail ? 1 : 0;
}
