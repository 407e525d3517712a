// tsq_host_test.cpp — the reference's own SQL-level test cases, replayed through the C++ host executors
// (tsq_host.hpp) on the GPU.  Each case cites the reference test that holds the expected rows; like testkit's
// `Sort().Check(testkit.Rows(...))` rows are rendered as strings ("<nil>" for NULL) and compared sorted.
// Run by tests/test_host_cpp_gpu.py; exit status 0 = every case passed.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <functional>
#include <map>
#include <sstream>
#include <unordered_map>

#include "tsq_host.hpp"

using namespace tsqhost;

static int g_fail = 0, g_pass = 0;
typedef std::vector<std::string> Rows;

static std::string cell(const Column& c, int64_t r) {
    if (c.IsNull(r)) return "<nil>";
    char buf[64];
    if (c.type == TSQ_BYTES) return c.GetString(r);
    switch (c.type) {
        case TSQ_I64: snprintf(buf, sizeof buf, "%lld", (long long)c.GetInt64(r)); break;
        case TSQ_U64: snprintf(buf, sizeof buf, "%llu", (unsigned long long)c.GetUint64(r)); break;
        case TSQ_F32: snprintf(buf, sizeof buf, "%g", (double)c.GetFloat32(r)); break;
        default: snprintf(buf, sizeof buf, "%.17g", c.GetFloat64(r)); break;
    }
    return buf;
}
static Rows render(const std::vector<Chunk>& chunks) {
    Rows out;
    for (auto& chk : chunks)
        for (int64_t r = 0; r < chk.NumRows(); r++) {
            std::string s;
            for (int c = 0; c < chk.NumCols(); c++) s += (c ? " " : "") + cell(chk.columns[c], r);
            out.push_back(s);
        }
    std::sort(out.begin(), out.end());
    return out;
}
static Rows render_in_order(const std::vector<Chunk>& chunks) {  // (StreamAggExec / SortExec: the order is part of the contract)
    Rows out;
    for (auto& chk : chunks)
        for (int64_t r = 0; r < chk.NumRows(); r++) {
            std::string s;
            for (int c = 0; c < chk.NumCols(); c++) s += (c ? " " : "") + cell(chk.columns[c], r);
            out.push_back(s);
        }
    return out;
}
static void expect(const char* name, Rows got, Rows want) {
    std::sort(want.begin(), want.end());
    if (got == want) { g_pass++; printf("PASS %s (%zu rows)\n", name, got.size()); return; }
    g_fail++;
    printf("FAIL %s\n  got :", name);
    for (auto& r : got) printf(" [%s]", r.c_str());
    printf("\n  want:");
    for (auto& r : want) printf(" [%s]", r.c_str());
    printf("\n");
}
static void expect_true(const char* name, bool ok) {
    if (ok) { g_pass++; printf("PASS %s\n", name); } else { g_fail++; printf("FAIL %s\n", name); }
}

// `insert into t values (...)`: NIL marks a NULL cell
static const int64_t NIL = INT64_MIN + 12345;
static Chunk table_i64(int ncols, std::initializer_list<int64_t> cells) {
    Chunk t(Schema((size_t)ncols, TSQ_I64));
    int c = 0;
    for (int64_t v : cells) {
        if (v == NIL) t.columns[c].AppendNull(); else t.columns[c].AppendInt64(v);
        c = (c + 1) % ncols;
    }
    return t;
}

// a table with mixed column types: cells are strings, "NULL" marks a NULL cell; ints / floats are parsed
static Chunk table_mixed(const Schema& types, std::initializer_list<const char*> cells) {
    Chunk t(types);
    size_t c = 0;
    for (const char* v : cells) {
        Column& col = t.columns[c];
        if (!strcmp(v, "NULL")) col.AppendNull();
        else if (col.type == TSQ_BYTES) col.AppendString(v);
        else if (col.type == TSQ_F32) col.AppendFloat32((float)atof(v));
        else if (col.type == TSQ_F64) col.AppendFloat64(atof(v));
        else col.AppendInt64(atoll(v));
        c = (c + 1) % types.size();
    }
    return t;
}

int main() {
    if (tsq_device_count() <= 0) { printf("no HIP device: the host executors have no CPU fallback\n"); return 2; }
    Context ctx(0);
    ctx.Reserve(512ll << 20);  // what the host process does once at start-up: every operator below works out of this slab

    // ---- executor/join_test.go:134-160: t = (1,1),(2,2),(3,3); t1 = (1,2),(1,3),(1,4),(3,4),(4,5)
    {
        Chunk t = table_i64(2, {1, 1, 2, 2, 3, 3}), t1 = table_i64(2, {1, 2, 1, 3, 1, 4, 3, 4, 4, 5});
        for (int inner = 0; inner < 2; inner++) {  // either child may be the build side (exhaust_physical_plans.go:247-275)
            MockDataSource l(&ctx, t), r(&ctx, t1);
            HashJoinExec j(&ctx, &l, &r, {0}, {0}, InnerJoin, inner);
            expect("join_test.go:134-146 t join t1 on t.c1 = t1.c1", render(Drain(&j)), {"1 1 1 2", "1 1 1 3", "1 1 1 4", "3 3 3 4"});
        }
        {
            MockDataSource l(&ctx, t), r(&ctx, t1);
            HashJoinExec j(&ctx, &l, &r, {0}, {0}, RightOuterJoin, 0);
            expect("join_test.go:148-160 t right outer join t1", render(Drain(&j)), {"1 1 1 2", "1 1 1 3", "1 1 1 4", "3 3 3 4", "<nil> <nil> 4 5"});
        }
        {
            MockDataSource l(&ctx, t), r(&ctx, t1);
            HashJoinExec j(&ctx, &l, &r, {0}, {0}, LeftOuterJoin, 1);
            expect("join_test.go:74-83 left outer join pads NULLs", render(Drain(&j)), {"1 1 1 2", "1 1 1 3", "1 1 1 4", "2 2 <nil> <nil>", "3 3 3 4"});
        }
    }
    // ---- join_test.go:101-104: duplicate keys, 3 x 3 rows of (1) -> nine "1 1"
    {
        Chunk a = table_i64(1, {1, 1, 1}), b = table_i64(1, {1, 1, 1});
        MockDataSource l(&ctx, a), r(&ctx, b);
        HashJoinExec j(&ctx, &l, &r, {0}, {0}, InnerJoin, 1);
        expect("join_test.go:101-104 3x3 duplicates", render(Drain(&j)), Rows(9, "1 1"));
    }
    // ---- join_test.go:112-116: 1..7 join 1..7 on a.c1 = b.c1 and a.c1 + b.c1 > 5 -> 3..7
    {
        Chunk a = table_i64(1, {1, 2, 3, 4, 5, 6, 7}), b = table_i64(1, {1, 2, 3, 4, 5, 6, 7});
        MockDataSource l(&ctx, a), r(&ctx, b);
        HashJoinExec j(&ctx, &l, &r, {0}, {0}, InnerJoin, 1, {Func("gt", {Func("plus", {Col(0, TSQ_I64), Col(1, TSQ_I64)}), Int(5)})});
        expect("join_test.go:112-116 other condition a.c1+b.c1>5", render(Drain(&j)), {"3 3", "4 4", "5 5", "6 6", "7 7"});
    }
    // ---- inline projection (planner/core/rule_column_pruning.go): the parent reads a.k, a.v, b.v only — the join may leave b.k unmaterialised
    {
        Chunk a = table_i64(2, {1, 10, 2, 20, 3, 30}), b = table_i64(2, {2, 200, 3, 300, 4, 400});
        MockDataSource l(&ctx, a), r(&ctx, b);
        HashJoinExec j(&ctx, &l, &r, {0}, {0}, InnerJoin, 1);
        j.SetUsedColumns({1, 1, 0, 1});
        Rows got;
        for (auto& chk : Drain(&j))
            for (int64_t i = 0; i < chk.NumRows(); i++) got.push_back(cell(chk.columns[0], i) + " " + cell(chk.columns[1], i) + " " + cell(chk.columns[3], i));
        std::sort(got.begin(), got.end());
        expect("used columns {a.k, a.v, b.v} of an inner join", got, {"2 20 200", "3 30 300"});
        expect_true("no division-by-zero warnings without conditions", j.DivisionByZeroWarnings() == 0);
    }
    // ---- NULL keys never join (hash_table.go:161-163, join.go:344); the outer side keeps them
    {
        Chunk a = table_i64(2, {NIL, 1, 2, 2, NIL, 3}), b = table_i64(2, {NIL, 10, 2, 20});
        MockDataSource l(&ctx, a), r(&ctx, b);
        HashJoinExec j(&ctx, &l, &r, {0}, {0}, LeftOuterJoin, 1);
        expect("NULL keys: left outer join", render(Drain(&j)), {"<nil> 1 <nil> <nil>", "<nil> 3 <nil> <nil>", "2 2 2 20"});
    }
    // ---- join_test.go:181-182: 100 x 100 duplicate join, LIMIT 1 closes the executor early (TestJoinLeak shape)
    {
        Chunk a(Schema{TSQ_I64}), b(Schema{TSQ_I64});
        for (int i = 0; i < 100; i++) { a.columns[0].AppendInt64(1); b.columns[0].AppendInt64(1); }
        MockDataSource l(&ctx, a), r(&ctx, b);
        HashJoinExec j(&ctx, &l, &r, {0}, {0}, InnerJoin, 1);
        j.Open();
        Chunk req(j.schema(), 1);  // LIMIT 1: the parent asks for one row
        j.Next(&req);
        const bool one = req.NumRows() == 1 && cell(req.columns[0], 0) == "1";
        j.Close();  // must not hang or leak with 9999 rows still undelivered
        expect_true("join_test.go:181-182 early Close after LIMIT 1", one);
        MockDataSource l2(&ctx, a), r2(&ctx, b);
        HashJoinExec j2(&ctx, &l2, &r2, {0}, {0}, InnerJoin, 1);
        int64_t total = 0;
        for (auto& c : Drain(&j2)) total += c.NumRows();
        expect_true("join_test.go:181 100x100 -> 10000 rows in <=1024-row chunks", total == 10000);
    }
    // ---- executor/aggregate_test.go:58-68: count by group; empty table -> no rows
    {
        Chunk empty(Schema{TSQ_I64, TSQ_I64});
        MockDataSource src(&ctx, empty);
        HashAggExec agg(&ctx, &src, {0}, {{TSQ_AGG_COUNT, 1, TSQ_I64}});
        expect("aggregate_test.go:58-59 empty input with GROUP BY", render(Drain(&agg)), {});
        Chunk t = table_i64(2, {1, 1, 2, 1, 3, 1, 4, 1, 4, 2, 4, 3});  // groups 1,2,3 -> 1 row each, 4 -> 3 rows
        MockDataSource src2(&ctx, t);
        HashAggExec agg2(&ctx, &src2, {0}, {{TSQ_AGG_COUNT, 1, TSQ_I64}});
        expect("aggregate_test.go:60-68 count(c) group by", render(Drain(&agg2)), {"1", "1", "1", "3"});
    }
    // ---- empty input WITHOUT group by: one row of defaults, COUNT -> 0, others NULL (builder.go:517-539, aggregate.go:572-574)
    {
        Chunk empty(Schema{TSQ_I64});
        MockDataSource src(&ctx, empty);
        HashAggExec agg(&ctx, &src, {}, {{TSQ_AGG_COUNT, -1, TSQ_I64}, {TSQ_AGG_SUM, 0, TSQ_I64}, {TSQ_AGG_MAX, 0, TSQ_I64}});
        expect("aggregate.go:572-574 default row on empty input", render(Drain(&agg)), {"0 <nil> <nil>"});
    }
    // ---- aggfuncs: SUM / AVG(int) = integer division (func_avg_test.go:23-24: expects 2), MIN/MAX, NULL args skipped
    {
        Chunk t = table_i64(2, {1, 0, 1, 1, 1, 2, 1, 3, 1, 4, 2, NIL, 2, NIL, NIL, 7});
        MockDataSource src(&ctx, t);
        HashAggExec agg(&ctx, &src, {0}, {{TSQ_AGG_FIRSTROW, 0, TSQ_I64}, {TSQ_AGG_COUNT, 1, TSQ_I64}, {TSQ_AGG_SUM, 1, TSQ_I64}, {TSQ_AGG_AVG, 1, TSQ_I64},
                                         {TSQ_AGG_MAX, 1, TSQ_I64}, {TSQ_AGG_MIN, 1, TSQ_I64}});
        expect("func_{count,sum,avg,max_min}_test.go: 0..4 -> 5 10 2 4 0; all-NULL group; NULL group", render(Drain(&agg)),
               {"1 5 10 2 4 0", "2 0 <nil> <nil> <nil> <nil>", "<nil> 1 7 7 7 7"});
    }
    // ---- StreamAggExec: the same cases on key-ordered input, groups IN INPUT ORDER (NULL group first: the order the child delivers)
    {
        Chunk t = table_i64(2, {NIL, 7, 1, 0, 1, 1, 1, 2, 1, 3, 1, 4, 2, NIL, 2, NIL});
        MockDataSource src(&ctx, t);
        StreamAggExec agg(&ctx, &src, {0}, {{TSQ_AGG_FIRSTROW, 0, TSQ_I64}, {TSQ_AGG_COUNT, 1, TSQ_I64}, {TSQ_AGG_SUM, 1, TSQ_I64}, {TSQ_AGG_AVG, 1, TSQ_I64},
                                           {TSQ_AGG_MAX, 1, TSQ_I64}, {TSQ_AGG_MIN, 1, TSQ_I64}});
        Rows got = render_in_order(Drain(&agg));
        expect_true("StreamAggExec (planner/core/cbo_test.go:200-212 StreamAgg): func_*_test.go values, groups in input order",
                    got == Rows{"<nil> 1 7 7 7 7", "1 5 10 2 4 0", "2 0 <nil> <nil> <nil> <nil>"});
        Chunk e(Schema{TSQ_I64});
        MockDataSource empty(&ctx, e);
        StreamAggExec agg0(&ctx, &empty, {}, {{TSQ_AGG_COUNT, -1, TSQ_I64}, {TSQ_AGG_SUM, 0, TSQ_I64}});
        expect("StreamAggExec: aggregate.go:572-574 default row on empty input", render(Drain(&agg0)), {"0 <nil>"});
    }
    // ---- SUM(int64) overflow is an error, not a wrap (func_sum.go:133-137, types/overflow.go:33-40)
    {
        Chunk t = table_i64(1, {INT64_MAX, 1});
        MockDataSource src(&ctx, t);
        HashAggExec agg(&ctx, &src, {}, {{TSQ_AGG_SUM, 0, TSQ_I64}});
        bool overflow = false;
        try { Drain(&agg); } catch (const Error& e) { overflow = e.IsOverflow(); }
        expect_true("func_sum.go:133-137 BIGINT overflow -> types.ErrOverflow", overflow);
    }
    // ---- Selection + Projection: select c1 + 1, c2 / 0 ... where c1 > 1 ; x/0 -> NULL + warning (errors.go:65-77)
    {
        Chunk t(Schema{TSQ_I64, TSQ_F64});
        for (int i = 1; i <= 5; i++) { t.columns[0].AppendInt64(i); t.columns[1].AppendFloat64(i * 0.5); }
        MockDataSource src(&ctx, t);
        SelectionExec sel(&ctx, &src, {Func("gt", {Col(0, TSQ_I64), Int(1)}), Func("ne", {Col(0, TSQ_I64), Int(4)})});
        ProjectionExec proj(&ctx, &sel, {Func("plus", {Col(0, TSQ_I64), Int(1)}), Func("div", {Col(1, TSQ_F64), Real(0.0)}), Func("mul", {Col(1, TSQ_F64), Real(2.0)})});
        proj.Open();
        Chunk req(proj.schema(), 1024);
        proj.Next(&req);
        std::vector<Chunk> one;
        one.push_back(req);
        const int64_t warns = proj.DivisionByZeroWarnings();
        proj.Close();
        expect("selection c1>1 and c1<>4; projection c1+1, c2/0, c2*2", render(one), {"3 <nil> 2", "4 <nil> 3", "6 <nil> 5"});
        expect_true("errors.go:65-77 division by zero -> NULL + one warning per row", warns == 3);
    }
    // ---- a projection that computes STRINGS: select if(c > 2, s, 'small'), ifnull(s, 'none') — builtinIfStringSig /
    //      builtinIfNullStringSig.vecEvalString (builtin_control_vec_generated.go:209, :81) through tsq_expr_eval_str
    {
        Chunk t(Schema{TSQ_I64, TSQ_BYTES});
        const char* names[5] = {"one", "two", nullptr, "four", ""};
        for (int i = 1; i <= 5; i++) {
            t.columns[0].AppendInt64(i);
            if (names[i - 1]) t.columns[1].AppendString(names[i - 1]);
            else t.columns[1].AppendNull();
        }
        MockDataSource src(&ctx, t);
        ProjectionExec proj(&ctx, &src, {Col(0, TSQ_I64), Func("if", {Func("gt", {Col(0, TSQ_I64), Int(2)}), Col(1, TSQ_BYTES), Str("small")}),
                                         Func("ifnull", {Col(1, TSQ_BYTES), Str("none")})});
        expect("projection c, if(c > 2, s, 'small'), ifnull(s, 'none')", render(Drain(&proj)),
               {"1 small one", "2 small two", "3 <nil> none", "4 four four", "5  "});
    }
    // ---- arithmetic overflow aborts the statement (builtin_arithmetic_vec.go:481-495 plusSS)
    {
        Chunk t = table_i64(1, {1, INT64_MAX});
        MockDataSource src(&ctx, t);
        ProjectionExec proj(&ctx, &src, {Func("plus", {Col(0, TSQ_I64), Int(1)})});
        bool overflow = false;
        try { Drain(&proj); } catch (const Error& e) { overflow = e.code == TSQ_ERR_OVERFLOW_BIGINT; }
        expect_true("builtin_arithmetic_vec.go:481-495 BIGINT overflow in c+1", overflow);
    }
    // ---- unsupported plans are refused at construction so that the Go operator can run instead
    {
        bool refused = false;
        try { Func("plus", {Col(0, TSQ_I64), Real(1.0)}); } catch (const Error& e) { refused = e.IsUnsupported(); }
        expect_true("mixed int/real arithmetic -> TSQ_ERR_UNSUPPORTED (no CAST in TinySQL)", refused);
    }
    // ---- a plan: select b.k, count(*), sum(a.v) from a join b on a.k = b.k where a.v > 10 group by b.k — vs a std::map restatement
    {
        const int na = 50000, nb = 3000;
        Chunk a(Schema{TSQ_I64, TSQ_I64}), b(Schema{TSQ_I64});
        uint64_t x = 88172645463325252ULL;
        auto rnd = [&]() { x ^= x << 13; x ^= x >> 7; x ^= x << 17; return x; };
        std::unordered_map<int64_t, int> bcount;
        for (int i = 0; i < nb; i++) { int64_t k = (int64_t)(rnd() % 2000); b.columns[0].AppendInt64(k); bcount[k]++; }
        std::map<int64_t, std::pair<int64_t, int64_t>> want;
        for (int i = 0; i < na; i++) {
            int64_t k = (int64_t)(rnd() % 2500), v = (int64_t)(rnd() % 100);
            if (rnd() % 50 == 0) { a.columns[0].AppendNull(); a.columns[1].AppendInt64(v); continue; }
            a.columns[0].AppendInt64(k);
            a.columns[1].AppendInt64(v);
            auto it = bcount.find(k);
            if (v > 10 && it != bcount.end()) { want[k].first += it->second; want[k].second += v * it->second; }
        }
        Rows wr;
        for (auto& kv : want) wr.push_back(std::to_string(kv.first) + " " + std::to_string(kv.second.first) + " " + std::to_string(kv.second.second));
        MockDataSource sa(&ctx, a), sb(&ctx, b);
        SelectionExec sel(&ctx, &sa, {Func("gt", {Col(1, TSQ_I64), Int(10)})});
        HashJoinExec j(&ctx, &sel, &sb, {0}, {0}, InnerJoin, 1);  // output: a.k a.v b.k
        HashAggExec agg(&ctx, &j, {2}, {{TSQ_AGG_FIRSTROW, 2, TSQ_I64}, {TSQ_AGG_COUNT, -1, TSQ_I64}, {TSQ_AGG_SUM, 1, TSQ_I64}});
        expect_true("Selection -> HashJoin -> HashAgg over 49 + 3 chunks equals a std::map restatement", render(Drain(&agg)) == [&] { std::sort(wr.begin(), wr.end()); return wr; }());
    }
    // ---- executor/union_scan_test.go:31-35: ORDER BY rows (ordered comparison: no sorting of the rendered rows here)
    {
        auto ordered = [](const std::vector<Chunk>& chunks) {
            Rows out;
            for (auto& chk : chunks)
                for (int64_t r = 0; r < chk.NumRows(); r++) {
                    std::string s;
                    for (int c = 0; c < chk.NumCols(); c++) s += (c ? " " : "") + cell(chk.columns[c], r);
                    out.push_back(s);
                }
            return out;
        };
        Chunk t = table_i64(2, {1, 5, 2, 3, 3, 4, 4, 8, 6, 8, 7, 6});
        {
            MockDataSource src(&ctx, t);
            SortExec so(&ctx, &src, {{0, true}});
            expect_true("union_scan_test.go:33 order by a desc", ordered(Drain(&so)) == Rows{"7 6", "6 8", "4 8", "3 4", "2 3", "1 5"});
        }
        {
            MockDataSource src(&ctx, t);
            SortExec so(&ctx, &src, {{1, false}, {0, false}});
            expect_true("union_scan_test.go:34 order by b, a", ordered(Drain(&so)) == Rows{"2 3", "3 4", "1 5", "7 6", "4 8", "6 8"});
        }
        {
            MockDataSource src(&ctx, t);
            SortExec so(&ctx, &src, {{1, true}, {0, true}});
            expect_true("union_scan_test.go:35 order by b desc, a desc", ordered(Drain(&so)) == Rows{"6 8", "4 8", "7 6", "1 5", "3 4", "2 3"});
        }
        {
            MockDataSource src(&ctx, t);
            TopNExec top(&ctx, &src, {{1, true}, {0, true}}, 2, 1);
            expect_true("distsql_test.go:158 shape: order by b desc limit 2,1", ordered(Drain(&top)) == Rows{"7 6"});
        }
        // ---- executor/merge_join_test.go:245-258, 276-279, 316-321: rows IN ORDER through MergeJoinExec
        {
            Chunk mt = table_i64(2, {1, 1, 2, 2}), mt1 = table_i64(2, {2, 3, 4, 4});
            MockDataSource l(&ctx, mt), r(&ctx, mt1);
            MergeJoinExec mj(&ctx, &l, &r, {0}, {0}, LeftOuterJoin, 1, {}, {Func("ne", {Col(0, TSQ_I64), Int(1)})});
            expect_true("merge_join_test.go:257-258 left outer join on t.c1 = t1.c1 and t.c1 != 1", ordered(Drain(&mj)) == Rows{"1 1 <nil> <nil>", "2 2 2 3"});
            MockDataSource l2(&ctx, mt1), r2(&ctx, mt);
            MergeJoinExec mj2(&ctx, &l2, &r2, {0}, {0}, RightOuterJoin, 0);
            expect_true("merge_join_test.go:251 t1 right outer join t", ordered(Drain(&mj2)) == Rows{"<nil> <nil> 1 1", "2 3 2 2"});
            Chunk tt = table_i64(2, {1, 1, 1, 2, 1, 3, 1, 4}), ss = table_i64(1, {1});
            MockDataSource l3(&ctx, tt), r3(&ctx, ss);
            MergeJoinExec mj3(&ctx, &l3, &r3, {0}, {0}, InnerJoin, 1);
            expect_true("merge_join_test.go:316-321 t join s: four rows in t order", ordered(Drain(&mj3)) == Rows{"1 1 1", "1 2 1", "1 3 1", "1 4 1"});
            Chunk o = table_i64(2, {NIL, 0, 1, 1, 1, 2, 3, 3, 5, 4, 5, 5, 9, 6}), in = table_i64(2, {NIL, 10, NIL, 11, 1, 12, 2, 13, 5, 14, 5, 15, 7, 16});
            MockDataSource l4(&ctx, o), r4(&ctx, in);
            MergeJoinExec mj4(&ctx, &l4, &r4, {0}, {0}, LeftOuterJoin, 1);
            expect_true("merge_join.go:148-156,257-310 NULL keys, group walk, outer rows in order",
                        ordered(Drain(&mj4)) == Rows{"<nil> 0 <nil> <nil>", "1 1 1 12", "1 2 1 12", "3 3 <nil> <nil>", "5 4 5 14", "5 4 5 15", "5 5 5 14", "5 5 5 15", "9 6 <nil> <nil>"});
        }
        {   // NULL is the smallest value (compare.go:48-56); executor_test.go:504: limit 18446744073709551615 = no limit
            Chunk n = table_i64(1, {3, NIL, -7, NIL, 0});
            MockDataSource src(&ctx, n);
            TopNExec top(&ctx, &src, {{0, false}}, 0, UINT64_MAX);
            expect_true("compare.go:48-56 NULLs first; executor_test.go:504 max limit", ordered(Drain(&top)) == Rows{"<nil>", "<nil>", "-7", "0", "3"});
        }
    }
    // ---- strings (varchar columns travel as TSQ_BYTES: offsets + data, util/chunk/column.go:28-34)
    {
        {   // join_test.go:184-191: user(id, name) left join aa left join bb where bb.id < 10 -> "1 a"
            Chunk user = table_mixed({TSQ_I64, TSQ_BYTES}, {"1", "a", "2", "b"}), aa = table_i64(1, {1}), bb = table_i64(1, {1});
            MockDataSource u(&ctx, user), a(&ctx, aa), b(&ctx, bb);
            HashJoinExec j1(&ctx, &u, &a, {0}, {0}, LeftOuterJoin, 1);
            HashJoinExec j2(&ctx, &j1, &b, {2}, {0}, LeftOuterJoin, 1);
            SelectionExec sel(&ctx, &j2, {Func("lt", {Col(3, TSQ_I64), Int(10)})});
            expect("join_test.go:184-191 string payload through two outer joins and a filter", render(Drain(&sel)), {"1 a 1 1"});
        }
        {   // join_test.go:336-341 TestIssue5255: t1(a int, b varchar, c float) join t2(a) -> "1 2017-11-29 2.2 1"
            Chunk t1 = table_mixed({TSQ_I64, TSQ_BYTES, TSQ_F32}, {"1", "2017-11-29", "2.2"}), t2 = table_i64(1, {1});
            for (int inner = 0; inner < 2; inner++) {
                MockDataSource l(&ctx, t1), r(&ctx, t2);
                HashJoinExec j(&ctx, &l, &r, {0}, {0}, InnerJoin, inner);
                expect("join_test.go:336-341 varchar + float payload", render(Drain(&j)), {"1 2017-11-29 2.2 1"});
            }
        }
        {   // join_test.go:325-329: ... where t2.name = 'xxx' (builtinEQStringSig, builtin_compare_vec_generated.go:393)
            Chunk t2 = table_mixed({TSQ_I64, TSQ_BYTES, TSQ_BYTES}, {"1", "xxx", "2003-06-09 10:51:26", "2", "xxy", "never", "3", "NULL", "never"});
            MockDataSource src(&ctx, t2);
            SelectionExec sel(&ctx, &src, {Func("eq", {Col(1, TSQ_BYTES), Str("xxx")})});
            expect("join_test.go:325-329 filter name = 'xxx'", render(Drain(&sel)), {"1 xxx 2003-06-09 10:51:26"});
        }
        {   // string join keys: equal bytes join, NULL never joins, '' is a value (codec.go:233-235, 363-382)
            Chunk l = table_mixed({TSQ_BYTES, TSQ_I64}, {"ab", "1", "", "2", "NULL", "3", "abc", "4", "ab", "5"});
            Chunk r = table_mixed({TSQ_BYTES, TSQ_I64}, {"ab", "10", "", "20", "NULL", "30", "b", "40"});
            MockDataSource ls(&ctx, l), rs(&ctx, r);
            HashJoinExec j(&ctx, &ls, &rs, {0}, {0}, LeftOuterJoin, 1);
            expect("string join key, left outer", render(Drain(&j)),
                   {"ab 1 ab 10", "ab 5 ab 10", " 2  20", "<nil> 3 <nil> <nil>", "abc 4 <nil> <nil>"});
        }
        {   // GROUP BY a varchar: firstrow(name), count(*), max / min of a varchar (func_first_row.go:193-230, func_max_min.go:312-378)
            Chunk t = table_mixed({TSQ_BYTES, TSQ_BYTES, TSQ_I64},
                                  {"x", "pear", "1", "y", "fig", "2", "x", "apple", "3", "NULL", "kiwi", "4", "x", "NULL", "5", "", "plum", "6", "NULL", "NULL", "7"});
            MockDataSource src(&ctx, t);
            HashAggExec agg(&ctx, &src, {0},
                            {{TSQ_AGG_FIRSTROW, 0, TSQ_BYTES}, {TSQ_AGG_COUNT, -1, TSQ_I64},
                             {TSQ_AGG_MAX, 1, TSQ_BYTES}, {TSQ_AGG_MIN, 1, TSQ_BYTES},
                             {TSQ_AGG_SUM, 2, TSQ_I64}});
            expect("group by varchar: firstrow / count / max / min of strings", render(Drain(&agg)),
                   {"x 3 pear apple 9", "y 1 fig fig 2", "<nil> 2 kiwi kiwi 11", " 1 plum plum 6"});
        }
        {   // func_max_min_test.go:31,37,50,56: max / min of "0".."4" without GROUP BY; empty input -> NULL (aggregate.go:572-574)
            Chunk t = table_mixed({TSQ_BYTES}, {"0", "1", "2", "3", "4", "NULL"});
            MockDataSource src(&ctx, t);
            HashAggExec agg(&ctx, &src, {}, {{TSQ_AGG_MAX, 0, TSQ_BYTES}, {TSQ_AGG_MIN, 0, TSQ_BYTES},
                                             {TSQ_AGG_COUNT, 0, TSQ_BYTES}});
            expect("func_max_min_test.go:31,37 max/min of strings", render(Drain(&agg)), {"4 0 5"});
            Chunk none(Schema{TSQ_BYTES});
            MockDataSource empty(&ctx, none);
            HashAggExec agg0(&ctx, &empty, {}, {{TSQ_AGG_MAX, 0, TSQ_BYTES}, {TSQ_AGG_COUNT, 0, TSQ_BYTES}});
            expect("aggregate.go:572-574 empty input: max(string) NULL, count 0", render(Drain(&agg0)), {"<nil> 0"});
        }
    }
    // ---- the storage side (round 2): tablecodec's record keys, string payload through TopNExec, mocktikv's executor chain
    {
        // tablecodec_test.go:42-53, 111-135: EncodeRowKeyWithHandle / DecodeRowKey; RecordRowKeyLen = 19
        const std::vector<uint8_t> key = EncodeRowKeysWithHandles(&ctx, 1, {2});
        const uint8_t want[19] = {'t', 0x80, 0, 0, 0, 0, 0, 0, 1, '_', 'r', 0x80, 0, 0, 0, 0, 0, 0, 2};
        expect_true("tablecodec_test.go:42-53 EncodeRowKeyWithHandle(1, 2)", key.size() == 19 && !memcmp(key.data(), want, 19));
        const std::vector<int64_t> hs = {0, -1, 4294967295LL, INT64_MAX, INT64_MIN};
        expect_true("tablecodec_test.go:111-135 DecodeRowKey of encoded handles", DecodeRowKeys(&ctx, EncodeRowKeysWithHandles(&ctx, 55, hs)) == hs);
        bool threw = false;
        try { DecodeRowKeys(&ctx, std::vector<uint8_t>(19, 'x')); } catch (const Error& e) { threw = std::string(e.what()).find("invalid key") != std::string::npos; }
        expect_true("tablecodec.go:235-242 DecodeRowKey: invalid key", threw);
    }
    {   // ORDER BY v LIMIT with a varchar payload column (sort.go:146-318; the string cells travel with their rows)
        Chunk t = table_mixed({TSQ_BYTES, TSQ_I64}, {"pear", "3", "NULL", "1", "", "2", "fig", "5", "apple", "4"});
        MockDataSource src(&ctx, t);
        TopNExec top(&ctx, &src, {{1, true}}, 1, 3);
        std::vector<Chunk> got = Drain(&top);
        Rows ordered;
        for (auto& chk : got) for (int64_t r = 0; r < chk.NumRows(); r++) ordered.push_back(cell(chk.columns[0], r) + " " + cell(chk.columns[1], r));
        expect_true("TopN with a string payload: rows 1..3 of ORDER BY v DESC", ordered == Rows{"apple 4", "pear 3", " 2"});
    }
    {
        // a table (k, v, name) with an int handle; stored rows assembled like row.toBytes (util/rowcodec/row.go:80-99): small ids, one-byte ints
        struct R { int64_t h; int k, v; const char* name; };
        const R rows[] = {{1, 1, 10, "a"}, {2, 2, 20, "bb"}, {3, 1, 30, nullptr}, {4, 2, 5, ""}};
        mocktikv::Pairs pairs;
        std::vector<int64_t> handles;
        pairs.valueOffsets.push_back(0);
        for (const R& r : rows) {
            handles.push_back(r.h);
            std::vector<uint8_t>& b = pairs.values;
            const size_t nameLen = r.name ? strlen(r.name) : 0;
            const int notNull = r.name ? 3 : 2;
            b.insert(b.end(), {128, 0, (uint8_t)notNull, 0, (uint8_t)(3 - notNull), 0});
            b.insert(b.end(), {1, 2, 3});  // the not-null ids sorted, then the null ids: id 3 comes last either way
            const uint16_t ends[3] = {1, 2, (uint16_t)(2 + nameLen)};
            for (int i = 0; i < notNull; i++) { b.push_back((uint8_t)ends[i]); b.push_back((uint8_t)(ends[i] >> 8)); }
            b.push_back((uint8_t)r.k);
            b.push_back((uint8_t)r.v);
            if (r.name) b.insert(b.end(), r.name, r.name + nameLen);
            pairs.valueOffsets.push_back((int64_t)b.size());
        }
        pairs.keys = EncodeRowKeysWithHandles(&ctx, 41, handles);
        const std::vector<mocktikv::ColInfo> cols = {{1, TSQ_I64, false}, {2, TSQ_I64, false}, {3, TSQ_BYTES, false}, {-1, TSQ_I64, true}};
        {   // tableScanExec -> limitExec -> response: the handle from the key, the string as a compact-bytes datum, 64 rows per chunk
            mocktikv::tableScanExec scan(&ctx, cols, &pairs);
            mocktikv::limitExec lim(&ctx, &scan, 3);
            const std::vector<std::string> chunks = mocktikv::fillUpData4SelectResponse(&ctx, &lim, {3, 0, 2});
            const std::string want = std::string("\x08\x02\x08\x02\x02\x02" "a", 7) + std::string("\x08\x04\x08\x04\x02\x04" "bb", 8) + std::string("\x08\x06\x08\x02\x00", 5);
            expect_true("executor.go:48-196,472-507 + cop_handler_dag.go:414-425: scan -> limit 3 -> RowsData", chunks.size() == 1 && chunks[0] == want);
        }
        {   // tableScanExec -> selectionExec (v > 7) -> hashAggExec (count(*), sum(v), avg(v) group by k): partial results, then the group-by value
            mocktikv::tableScanExec scan(&ctx, cols, &pairs);
            mocktikv::selectionExec sel(&ctx, &scan, {Func("gt", {Col(1, TSQ_I64), Int(7)})});
            mocktikv::hashAggExec agg(&ctx, &sel, {{TSQ_AGG_COUNT, -1}, {TSQ_AGG_SUM, 1}, {TSQ_AGG_AVG, 1}}, {0});
            expect("aggregate.go:78-116 scan -> selection -> hashAgg: [count, sum, avg.count, avg.sum, k]", render(Drain(&agg)), {"2 40 2 40 1", "1 20 1 20 2"});
        }
        {   // topNExec above the scan: ORDER BY v DESC LIMIT 2, the varchar column travels along
            mocktikv::tableScanExec scan(&ctx, cols, &pairs);
            mocktikv::topNExec top(&ctx, &scan, {{1, true}}, 2);
            expect("executor.go:392-470 scan -> topN", render(Drain(&top)), {"1 30 <nil> 3", "2 20 bb 2"});
        }
    }
    // ---- tablecodec_test.go:55-75 TestCutKeyNew: (1, "abc", 5.5) + handle 100 under table 4 / index 5, as an index scan's pair;
    //      the same row from a unique index (no handle in the key: the pair's value holds it)
    {
        auto be = [](std::vector<uint8_t>& b, uint64_t v) { for (int i = 7; i >= 0; i--) b.push_back((uint8_t)(v >> (8 * i))); };
        const uint64_t sign = 0x8000000000000000ull;
        mocktikv::IndexPairs ip;
        for (int withHandle = 1; withHandle >= 0; withHandle--) {
            ip.keyOffsets.push_back((int64_t)ip.keys.size());
            ip.valueOffsets.push_back((int64_t)ip.values.size());
            ip.keys.push_back('t'); be(ip.keys, 4 ^ sign); ip.keys.push_back('_'); ip.keys.push_back('i'); be(ip.keys, 5 ^ sign);  // EncodeIndexSeekKey
            ip.keys.push_back(3); be(ip.keys, 1 ^ sign);                                                                           // intFlag + EncodeInt(1)
            ip.keys.push_back(1); for (char ch : std::string("abc")) ip.keys.push_back((uint8_t)ch);                               // bytesFlag + EncodeBytes("abc")
            for (int i = 0; i < 5; i++) ip.keys.push_back(0);
            ip.keys.push_back(250);
            ip.keys.push_back(5); be(ip.keys, 0xC016000000000000ull);                                                              // floatFlag + EncodeFloat(5.5)
            if (withHandle) { ip.keys.push_back(3); be(ip.keys, 100 ^ sign); ip.values.push_back('0'); }
            else be(ip.values, 100);                                                                                                // the handle as the value
        }
        ip.keyOffsets.push_back((int64_t)ip.keys.size());
        ip.valueOffsets.push_back((int64_t)ip.values.size());
        mocktikv::indexScanExec scan(&ctx, {TSQ_I64, TSQ_BYTES, TSQ_F64, TSQ_I64}, 3, mocktikv::PrimaryKeyIsSigned, &ip);
        expect("tablecodec_test.go:55-75 + tablecodec.go:406-434: index pairs -> (1, abc, 5.5, handle 100)", render(Drain(&scan)), {"1 abc 5.5 100", "1 abc 5.5 100"});
        mocktikv::indexScanExec noPk(&ctx, {TSQ_I64, TSQ_BYTES, TSQ_F64}, 3, mocktikv::PrimaryKeyNotExists, &ip);
        expect("tablecodec.go:411-414 PrimaryKeyNotExists drops the handle", render(Drain(&noPk)), {"1 abc 5.5", "1 abc 5.5"});
    }
    // ---- util/chunk/codec_test.go:29-71 TestCodec: (NULL, i, "<i>.12345", "<i>.12345") x 10 through Encode / DecodeToChunk
    {
        const Schema colTypes = {TSQ_I64, TSQ_I64, TSQ_BYTES, TSQ_BYTES};
        Chunk oldChk(colTypes, 10);
        for (int i = 0; i < 10; i++) {
            const std::string str = std::to_string(i) + ".12345";
            oldChk.columns[0].AppendNull();
            oldChk.columns[1].AppendInt64(i);
            oldChk.columns[2].AppendString(str);
            oldChk.columns[3].AppendString(str);
        }
        Codec codec(&ctx, colTypes);
        const std::vector<uint8_t> buffer = codec.Encode(oldChk);
        expect_true("codec.go:50-76: 8+2+80 | 8+80 | 8+88+70 | 8+88+70 wire bytes", buffer.size() == 510 && buffer[0] == 10 && buffer[4] == 10 && buffer[94] == 0);
        Chunk newChk(colTypes, 10);
        const std::vector<uint8_t> remained = codec.DecodeToChunk(buffer, newChk);
        expect_true("codec_test.go:52-55 remained / NumCols / NumRows", remained.empty() && newChk.NumCols() == 4 && newChk.NumRows() == 10);
        Rows want;
        for (int i = 0; i < 10; i++) want.push_back("<nil> " + std::to_string(i) + " " + std::to_string(i) + ".12345 " + std::to_string(i) + ".12345");
        expect("codec_test.go:56-69 TestCodec rows", render({newChk}), want);
        // Decoder: 10 rows into chunks that ask for 3 rows -> 8 + 2 (multiples of 8, codec.go:259), then the rest by ReuseIntermChk
        Chunk interm(colTypes), part(colTypes, 3), rest(colTypes);
        Decoder dec(&ctx, &interm, colTypes);
        dec.Reset(buffer);
        dec.Decode(part);
        expect_true("codec.go:257-269 Decode takes a multiple of 8 rows", part.NumRows() == 8 && dec.RemainedRows() == 2 && !dec.IsFinished());
        dec.ReuseIntermChk(rest);
        expect_true("codec.go:291-308 ReuseIntermChk", dec.IsFinished() && rest.NumRows() == 2 && rest.columns[2].offsets[0] == 0);
        expect("Decoder: the two parts are the rows", render({part, rest}), want);
    }
    {   // every executor above is closed: its buffers went back to the slab they came from
        const Context::Arena a = ctx.ArenaStats();
        expect_true("tsq_ctx_reserve: the operators worked out of the arena and gave everything back", a.size == (512ll << 20) && a.peak > 0 && a.used == 0);
    }
    printf("%d passed, %d failed\n", g_pass, g_fail);
    return g_fail ? 1 : 0;
}
