// flatten.cpp -- turns the Builder's object tree into the flat arrays of rt_types.h.
//
// The reference traverses a pointer tree (BvhNode -> Box<dyn Hittable>, bvh.rs:159-187) in which
// HittableList / BoxObj / nested BvhNode are only grouping devices: the closest hit does not depend
// on them.  The flattener therefore dissolves all grouping, keeps every primitive with the chain of
// Translate / RotateY wrappers above it (transrot.rs), and builds
//   * one bottom-level BVH per distinct chain, over that chain's primitives in object space,
//   * a top-level BVH over the untransformed primitives and the instances,
// with a binned-SAH builder instead of the reference's random-axis median split (bvh.rs:96-122).
// Constant media (mediums.rs) are kept out of the BVH: each becomes a record holding its boundary
// primitives, which the kernels test after the surface hit is known.
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <limits>
#include "scene_builder.h"

namespace rt {
namespace {

const double kInf = std::numeric_limits<double>::infinity();

struct Box3 {
    double mn[3], mx[3];
    Box3() { for (int a = 0; a < 3; ++a) { mn[a] = kInf; mx[a] = -kInf; } }
    void grow(const Box3& o) { for (int a = 0; a < 3; ++a) { mn[a] = std::min(mn[a], o.mn[a]); mx[a] = std::max(mx[a], o.mx[a]); } }
    void grow_pt(double x, double y, double z) {
        double p[3] = {x, y, z};
        for (int a = 0; a < 3; ++a) { mn[a] = std::min(mn[a], p[a]); mx[a] = std::max(mx[a], p[a]); }
    }
    double area() const {
        double d[3] = {mx[0] - mn[0], mx[1] - mn[1], mx[2] - mn[2]};
        if (d[0] < 0 || d[1] < 0 || d[2] < 0) return 0.0;
        return 2.0 * (d[0] * d[1] + d[1] * d[2] + d[2] * d[0]);
    }
    bool valid() const { return mn[0] <= mx[0] && mn[1] <= mx[1] && mn[2] <= mx[2]; }
};

struct Op { int type; double a, b, c; };  // translate: offset xyz | rotate_y: cos, sin
struct PrimRef { int obj; int chain; };
struct MediumRef { int obj; int chain; std::vector<PrimRef> boundary; };
struct Item { Box3 box; double cen[3]; int cls; int ref; };  // cls 0..3 = primitive kind, kInstanceClass = instance
const int kInstanceClass = 7;

float f_down(double v) { float f = (float)v; if ((double)f > v) f = std::nextafterf(f, -std::numeric_limits<float>::infinity()); return f; }
float f_up(double v) { float f = (float)v; if ((double)f < v) f = std::nextafterf(f, std::numeric_limits<float>::infinity()); return f; }
float i2f(int32_t i) { float f; std::memcpy(&f, &i, 4); return f; }
rt_f4 f4(float x, float y, float z, float w) { rt_f4 r; r.x = x; r.y = y; r.z = z; r.w = w; return r; }
int ceil_log2(size_t n) { int l = 0; while (((size_t)1 << l) < n) ++l; return l; }

class Flattener {
public:
    Flattener(const Builder& b_, FlatScene& out_, std::string& err_) : b(b_), out(out_), err(err_) {}

    bool defer_top = false;   // leave the top level to the linear builder: emit its items instead of nodes

    bool run(HittableObj world) {
        if (!b.valid_obj(world)) return fail("flatten: bad world handle");
        chains.push_back(std::vector<Op>());
        chain_prims.push_back(std::vector<PrimRef>());
        if (!walk(world, 0, nullptr, 0)) return false;
        out = FlatScene();
        out.n_ids = b.next_id;
        emit_tables();

        // Boundary primitives of the media first (they are never referenced by the BVH).
        for (MediumRef& m : media) if (!emit_medium(m)) return false;

        // One bottom-level BVH + instance record per non-empty transformed chain.
        std::vector<Item> top;
        int max_blas = 0;
        for (size_t c = 1; c < chains.size(); ++c) {
            if (chain_prims[c].empty()) continue;
            std::vector<Item> items;
            for (const PrimRef& p : chain_prims[c]) items.push_back(prim_item(p));
            int depth = 0;
            int root = build_root(items, kBlasBudget, depth);
            max_blas = std::max(max_blas, depth);
            Box3 wb = chain_world_box(node_box(root), chains[c]);
            Item it; it.box = wb; it.cls = kInstanceClass; it.ref = emit_instance((int)c, root);
            for (int a = 0; a < 3; ++a) it.cen[a] = 0.5 * (wb.mn[a] + wb.mx[a]);
            top.push_back(it);
        }
        for (const PrimRef& p : chain_prims[0]) top.push_back(prim_item(p));
        out.blas_depth = max_blas;
        if (defer_top && top.size() >= 2) {
            // One item per primitive or instance: (box.min, child code), (box.max, 0).  The nodes above them are built later.
            Box3 wb;
            out.tlas_items.reserve(top.size() * 2);
            for (const Item& it : top) {
                int code = it.cls == kInstanceClass ? rt_leaf_instance(it.ref) : rt_leaf_prims(it.cls, emit_prim(it.ref), 1);
                out.tlas_items.push_back(f4(f_down(it.box.mn[0]), f_down(it.box.mn[1]), f_down(it.box.mn[2]), i2f(code)));
                out.tlas_items.push_back(f4(f_up(it.box.mx[0]), f_up(it.box.mx[1]), f_up(it.box.mx[2]), 0));
                wb.grow(it.box);
            }
            for (int a = 0; a < 3; ++a) { out.world_box[a] = wb.mn[a]; out.world_box[3 + a] = wb.mx[a]; }
            out.root = -1; out.tlas_depth = 0;
            out.max_stack = out.instances.empty() ? 0 : 1 + max_blas;   // the top level's depth is added when it exists
            emit_medium_locals(top);
            relabel_breadth_first();
            return err.empty();
        }
        if (!top.empty()) {
            int budget = RT_STACK_DEPTH - 2 - (out.instances.empty() ? 0 : max_blas + 1);
            if (budget < 4) return fail("flatten: instance BVH too deep for the traversal stack");
            int depth = 0;
            out.root = build_root(top, budget, depth);
            out.tlas_depth = depth;
            Box3 wb = node_box(out.root);
            for (int a = 0; a < 3; ++a) { out.world_box[a] = wb.mn[a]; out.world_box[3 + a] = wb.mx[a]; }
        }
        out.max_stack = out.tlas_depth + (out.instances.empty() ? 0 : 1 + max_blas);
        if (out.max_stack > RT_STACK_DEPTH) return fail("flatten: BVH deeper than the traversal stack");
        emit_medium_locals(top);
        relabel_breadth_first();
        if (!err.empty()) return false;
        return true;
    }

private:
    static const int kBlasBudget = 18;
    const Builder& b; FlatScene& out; std::string& err;
    std::vector<std::vector<Op>> chains;
    std::vector<std::vector<PrimRef>> chain_prims;
    std::vector<MediumRef> media;
    std::vector<int> chain_instance;  // chain -> instance index (for media boundaries), -1 if none yet
    std::vector<int> index_of_obj;    // object -> index of its record in its kind's array (emit_prim)

    bool fail(const std::string& m) { err = m; return false; }

    // ---- tree walk (dissolves grouping, accumulates transform chains)
    bool walk(int oi, int chain, std::vector<PrimRef>* sink, int depth) {
        if (depth > 4096) return fail("flatten: object graph too deep (cycle?)");
        const ObjNode& o = b.objects[oi];
        switch (o.type) {
        case OBJ_BOX:  // the six sides stay together as one box primitive
        case OBJ_SPHERE: case OBJ_MSPHERE: case OBJ_RECT: {
            PrimRef p = {oi, chain};
            if (sink) sink->push_back(p); else chain_prims[chain].push_back(p);
            return true;
        }
        case OBJ_LIST: case OBJ_BVH:
            for (int c : o.children) if (!walk(c, chain, sink, depth + 1)) return false;
            return true;
        case OBJ_TRANSLATE: case OBJ_ROTATE_Y: {
            std::vector<Op> ops = chains[chain];
            Op op; op.type = o.type == OBJ_TRANSLATE ? RT_OP_TRANSLATE : RT_OP_ROTATE_Y;
            if (o.type == OBJ_TRANSLATE) { op.a = o.p[0]; op.b = o.p[1]; op.c = o.p[2]; }
            else { op.a = o.p[2]; op.b = o.p[1]; op.c = 0; }  // cos, sin
            ops.push_back(op);
            chains.push_back(ops); chain_prims.push_back(std::vector<PrimRef>());
            return walk(o.children[0], (int)chains.size() - 1, sink, depth + 1);
        }
        case OBJ_MEDIUM: {
            if (sink) return fail("flatten: a ConstantMedium cannot be part of another medium's boundary");
            MediumRef m; m.obj = oi; m.chain = -1;
            if (!walk(o.children[0], chain, &m.boundary, depth + 1)) return false;
            if (m.boundary.empty()) return fail("No bounding box in bvh constructor");  // bvh.rs:140 (empty boundary has no box)
            m.chain = m.boundary[0].chain;
            for (const PrimRef& p : m.boundary)
                if (p.chain != m.chain) return fail("flatten: a medium boundary must sit under a single transform chain");
            media.push_back(m);
            return true;
        }
        }
        return fail("flatten: unknown object type");
    }

    // ---- bounds
    Box3 prim_box(const ObjNode& o) const {
        Box3 bx;
        if (o.type == OBJ_SPHERE) {  // objects.rs:241-244, with |r| so that a negative radius keeps a valid box
            double r = std::fabs(o.p[3]);
            bx.grow_pt(o.p[0] - r, o.p[1] - r, o.p[2] - r); bx.grow_pt(o.p[0] + r, o.p[1] + r, o.p[2] + r);
        } else if (o.type == OBJ_MSPHERE) {  // objects.rs:328-335 at (time0, time1) = (0, 1), main.rs:531
            double r = std::fabs(o.p[8]);
            for (double tm = 0.0; tm <= 1.0; tm += 1.0) {
                double s = (tm - o.p[6]) / (o.p[7] - o.p[6]);
                double c[3] = {o.p[0] + s * (o.p[3] - o.p[0]), o.p[1] + s * (o.p[4] - o.p[1]), o.p[2] + s * (o.p[5] - o.p[2])};
                bx.grow_pt(c[0] - r, c[1] - r, c[2] - r); bx.grow_pt(c[0] + r, c[1] + r, c[2] + r);
            }
        } else if (o.type == OBJ_BOX) {  // objects.rs:514-516
            bx.grow_pt(o.p[0], o.p[1], o.p[2]); bx.grow_pt(o.p[3], o.p[4], o.p[5]);
        } else {  // objects.rs:413-429
            const double off = 0.0001;
            double x0 = o.p[0], x1 = o.p[1], y0 = o.p[2], y1 = o.p[3], k = o.p[4];
            if (o.rect_type == Xy) { bx.grow_pt(x0, y0, k - off); bx.grow_pt(x1, y1, k + off); }
            else if (o.rect_type == Xz) { bx.grow_pt(x0, k - off, y0); bx.grow_pt(x1, k + off, y1); }
            else { bx.grow_pt(k - off, x0, y0); bx.grow_pt(k + off, x1, y1); }
        }
        return bx;
    }
    // Object-space box -> world box through the chain, innermost op first (transrot.rs:34-38, 60-82).
    static Box3 chain_world_box(Box3 bx, const std::vector<Op>& ops) {
        for (int k = (int)ops.size() - 1; k >= 0; --k) {
            const Op& op = ops[k];
            if (op.type == RT_OP_TRANSLATE) {
                bx.mn[0] += op.a; bx.mx[0] += op.a; bx.mn[1] += op.b; bx.mx[1] += op.b; bx.mn[2] += op.c; bx.mx[2] += op.c;
            } else {
                Box3 r;
                for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int q = 0; q < 2; ++q) {
                    double x = i ? bx.mx[0] : bx.mn[0], y = j ? bx.mx[1] : bx.mn[1], z = q ? bx.mx[2] : bx.mn[2];
                    r.grow_pt(op.a * x + op.b * z, y, -op.b * x + op.a * z);
                }
                bx = r;
            }
        }
        return bx;
    }
    Item prim_item(const PrimRef& p) const {
        const ObjNode& o = b.objects[p.obj];
        Item it; it.box = prim_box(o);
        it.cls = o.type == OBJ_SPHERE ? RT_KIND_SPHERE : (o.type == OBJ_MSPHERE ? RT_KIND_MSPHERE : (o.type == OBJ_BOX ? RT_KIND_BOX : RT_KIND_RECT));
        it.ref = p.obj;
        for (int a = 0; a < 3; ++a) it.cen[a] = 0.5 * (it.box.mn[a] + it.box.mx[a]);
        return it;
    }

    // ---- emitters
    int emit_prim(int oi) {  // returns the index inside the primitive's kind array
        const int idx = emit_prim_record(oi);
        if (index_of_obj.size() < b.objects.size()) index_of_obj.resize(b.objects.size(), -1);
        index_of_obj[(size_t)oi] = idx;
        return idx;
    }
    int emit_prim_record(int oi) {
        const ObjNode& o = b.objects[oi];
        rt_i2 meta; meta.x = o.mat; meta.y = o.id;
        if (o.type == OBJ_SPHERE) {
            out.spheres.push_back(f4((float)o.p[0], (float)o.p[1], (float)o.p[2], (float)o.p[3]));
            out.sphere_meta.push_back(meta); return (int)out.spheres.size() - 1;
        }
        if (o.type == OBJ_MSPHERE) {
            out.mspheres.push_back(f4((float)o.p[0], (float)o.p[1], (float)o.p[2], (float)o.p[8]));
            out.mspheres.push_back(f4((float)(o.p[3] - o.p[0]), (float)(o.p[4] - o.p[1]), (float)(o.p[5] - o.p[2]), (float)o.p[6]));
            out.mspheres.push_back(f4((float)(1.0 / (o.p[7] - o.p[6])), 0, 0, 0));
            out.msphere_meta.push_back(meta); return (int)out.msphere_meta.size() - 1;
        }
        if (o.type == OBJ_BOX) {
            out.boxes.push_back(f4((float)o.p[0], (float)o.p[1], (float)o.p[2], i2f(o.mat)));
            out.boxes.push_back(f4((float)o.p[3], (float)o.p[4], (float)o.p[5], i2f(o.id)));
            return (int)out.boxes.size() / 2 - 1;
        }
        out.rects.push_back(f4((float)o.p[0], (float)o.p[1], (float)o.p[2], (float)o.p[3]));
        out.rects.push_back(f4((float)o.p[4], i2f(o.rect_type), 0, 0));
        out.rect_meta.push_back(meta); return (int)out.rect_meta.size() - 1;
    }
    // The surfaces a ray can meet while it is still inside a medium (src/mediums.rs:39-87 scatters rays from points inside the
    // boundary; the reference then searches the whole world again, src/lib.rs:63).  For a medium whose boundary is one sphere or
    // one box in world space, every surface point inside the boundary belongs to a primitive that meets the boundary's solid:
    // when those are at most three world-space primitives, the medium record lists them (as one-primitive leaf codes in m4.z,
    // m4.w, m5.w, those that lie wholly inside the boundary first; m5.z = count | mask << 8, bit k of the mask: entry k lies
    // wholly inside).  A ray that starts inside the boundary and hits a flagged primitive anywhere has found its closest surface
    // hit among the list's entries: the segment up to that hit lies inside the (convex) boundary, so whatever it meets first is
    // inside the boundary too and is in the list -- the shading kernel can settle that ray's closest-hit search itself.  An
    // instance near the boundary, or more than three candidates, leave the list empty and every ray goes through the tree.
    void emit_medium_locals(const std::vector<Item>& top) {
        if (index_of_obj.size() < b.objects.size()) index_of_obj.resize(b.objects.size(), -1);
        // everything a ray can start or end on: the surfaces of the trees and the other media's boundaries
        Box3 all;
        for (const Item& it : top) all.grow(it.box);
        std::vector<Box3> medium_box(media.size());
        for (size_t mi = 0; mi < media.size(); ++mi) {
            const rt_f4* rec = &out.media[mi * RT_MEDIUM_F4];
            medium_box[mi].grow_pt(rec[0].x, rec[0].y, rec[0].z); medium_box[mi].grow_pt(rec[1].x, rec[1].y, rec[1].z);
        }
        for (size_t mi = 0; mi < media.size(); ++mi) {
            rt_f4* rec = &out.media[mi * RT_MEDIUM_F4];
            int32_t fast, has_inst;
            std::memcpy(&fast, &rec[5].x, 4); std::memcpy(&has_inst, &rec[5].y, 4);
            if (fast == 0 || has_inst) continue;
            const ObjNode& bo = b.objects[media[mi].boundary[0].obj];
            const Box3 bb = prim_box(bo);
            if (fast == 1 && all.valid()) {
                // A sphere boundary whose radius is at least twice the distance from its centre to the farthest corner of
                // everything else: rays of the scene cross it far behind their origin and far beyond their end (rt_medium_collision).
                Box3 rest = all;
                for (size_t mj = 0; mj < media.size(); ++mj) if (mj != mi) rest.grow(medium_box[mj]);
                double far2 = 0.0;
                for (int c = 0; c < 8; ++c) {
                    const double dx = ((c & 1) ? rest.mx[0] : rest.mn[0]) - bo.p[0], dy = ((c & 2) ? rest.mx[1] : rest.mn[1]) - bo.p[1],
                                 dz = ((c & 4) ? rest.mx[2] : rest.mn[2]) - bo.p[2];
                    far2 = std::max(far2, dx * dx + dy * dy + dz * dz);
                }
                if (std::fabs(bo.p[3]) >= 2.0 * std::sqrt(far2)) rec[5].y = i2f(has_inst | 2);
            }
            const double size = std::fabs(bb.mx[0] - bb.mn[0]) + std::fabs(bb.mx[1] - bb.mn[1]) + std::fabs(bb.mx[2] - bb.mn[2]);
            const double pad = 1e-3 * (1.0 + size);
            // does the item's box come within `pad` of the boundary's solid?  (sphere boundary: distance from its centre to the box)
            auto meets = [&](const Box3& x) {
                if (fast == 1) {
                    double d2 = 0.0;
                    for (int a = 0; a < 3; ++a) { const double c = bo.p[a], d = c < x.mn[a] ? x.mn[a] - c : (c > x.mx[a] ? c - x.mx[a] : 0.0); d2 += d * d; }
                    return std::sqrt(d2) <= std::fabs(bo.p[3]) + pad;
                }
                for (int a = 0; a < 3; ++a) if (!(x.mn[a] <= bb.mx[a] + pad && x.mx[a] >= bb.mn[a] - pad)) return false;
                return true;
            };
            int codes[3] = {0, 0, 0}, n = 0, mask = 0; bool ok = true;
            for (const Item& it : top) {
                if (!meets(it.box)) continue;
                if (it.cls == kInstanceClass || n == 3 || index_of_obj[(size_t)it.ref] < 0) { ok = false; break; }
                bool inside = true;    // every corner of the primitive's box within the boundary (a sphere against a sphere boundary: exactly)
                const ObjNode& po = b.objects[(size_t)it.ref];
                if (fast == 1) {
                    const double r = std::fabs(bo.p[3]) * (1.0 + 1e-6);
                    if (po.type == OBJ_SPHERE) {
                        const double dx = po.p[0] - bo.p[0], dy = po.p[1] - bo.p[1], dz = po.p[2] - bo.p[2];
                        inside = std::sqrt(dx * dx + dy * dy + dz * dz) + std::fabs(po.p[3]) <= r;
                    } else
                        for (int c = 0; c < 8 && inside; ++c) {
                            const double dx = ((c & 1) ? it.box.mx[0] : it.box.mn[0]) - bo.p[0], dy = ((c & 2) ? it.box.mx[1] : it.box.mn[1]) - bo.p[1],
                                         dz = ((c & 4) ? it.box.mx[2] : it.box.mn[2]) - bo.p[2];
                            inside = std::sqrt(dx * dx + dy * dy + dz * dz) <= r;
                        }
                } else
                    for (int a = 0; a < 3; ++a) {
                        const double tol = 1e-6 * (1.0 + std::fabs(bb.mn[a]) + std::fabs(bb.mx[a]));
                        inside = inside && it.box.mn[a] >= bb.mn[a] - tol && it.box.mx[a] <= bb.mx[a] + tol;
                    }
                codes[n] = rt_leaf_prims(it.cls, index_of_obj[(size_t)it.ref], 1);
                if (inside) mask |= 1 << n;
                ++n;
            }
            if (!ok || n == 0 || mask == 0) continue;
            // flagged entries first: the kernel learns whether any of them is hit before an unflagged, nearer hit can hide it
            int ordered[3] = {0, 0, 0}, k = 0, n_flagged = 0;
            for (int pass = 0; pass < 2; ++pass)
                for (int i = 0; i < n; ++i) if (((mask >> i) & 1) == (pass == 0 ? 1 : 0)) { ordered[k++] = codes[i]; if (pass == 0) ++n_flagged; }
            rec[4].z = i2f(ordered[0]); rec[4].w = i2f(ordered[1]); rec[5].w = i2f(ordered[2]);
            rec[5].z = i2f(n | (((1 << n_flagged) - 1) << 8));
        }
    }
    int emit_instance(int chain, int root) {
        // Collapse the chain into one affine map: object point = R * world + c.
        double cs = 1, sn = 0, c[3] = {0, 0, 0};
        for (const Op& op : chains[chain]) {  // outermost first
            if (op.type == RT_OP_TRANSLATE) { c[0] -= op.a; c[1] -= op.b; c[2] -= op.c; }
            else {  // p' = Ry p : x' = cos x - sin z ; z' = sin x + cos z   (transrot.rs:96-97)
                double x = op.a * c[0] - op.b * c[2], z = op.b * c[0] + op.a * c[2];
                c[0] = x; c[2] = z;
                double ncs = op.a * cs - op.b * sn, nsn = op.b * cs + op.a * sn;
                cs = ncs; sn = nsn;
            }
        }
        int first = (int)out.ops.size();
        for (const Op& op : chains[chain])
            out.ops.push_back(op.type == RT_OP_TRANSLATE ? f4((float)op.a, (float)op.b, (float)op.c, i2f(RT_OP_TRANSLATE))
                                                         : f4((float)op.a, (float)op.b, 0, i2f(RT_OP_ROTATE_Y)));
        out.instances.push_back(f4((float)cs, (float)sn, (float)c[0], (float)c[1]));
        out.instances.push_back(f4((float)c[2], i2f(root), i2f(first), i2f((int)chains[chain].size())));
        return (int)out.instances.size() / RT_INSTANCE_F4 - 1;
    }
    bool emit_medium(const MediumRef& m) {
        const ObjNode& o = b.objects[m.obj];
        Box3 bx;
        int first[4] = {(int)out.spheres.size(), (int)out.msphere_meta.size(), (int)out.rect_meta.size(), (int)out.boxes.size() / 2};
        int count[4] = {0, 0, 0, 0};
        for (int kind = 0; kind < 4; ++kind)
            for (const PrimRef& p : m.boundary) {
                Item it = prim_item(p);
                if (it.cls != kind) continue;
                emit_prim(p.obj); count[kind]++; bx.grow(it.box);
            }
        int inst = -1;
        if (m.chain > 0) { inst = emit_instance(m.chain, -1); bx = chain_world_box(bx, chains[m.chain]); }
        if (!(o.p[0] != 0.0)) return fail("constant_medium: density must be non-zero");
        // generous fp32 bounds: the box only culls, the boundary test decides
        double pad[3];
        for (int a = 0; a < 3; ++a) pad[a] = 1e-5 * (std::fabs(bx.mn[a]) + std::fabs(bx.mx[a])) + 1e-4;
        out.media.push_back(f4(f_down(bx.mn[0] - pad[0]), f_down(bx.mn[1] - pad[1]), f_down(bx.mn[2] - pad[2]), (float)(-1.0 / o.p[0])));
        out.media.push_back(f4(f_up(bx.mx[0] + pad[0]), f_up(bx.mx[1] + pad[1]), f_up(bx.mx[2] + pad[2]), i2f(o.mat)));
        out.media.push_back(f4(i2f(inst), i2f(first[0]), i2f(count[0]), i2f(first[1])));
        out.media.push_back(f4(i2f(count[1]), i2f(first[2]), i2f(count[2]), i2f(o.id)));
        out.media.push_back(f4(i2f(first[3]), i2f(count[3]), 0, 0));
        // A boundary that is one sphere or one box (every shipped scene) is repeated inline, with its instance transform,
        // so that the medium test needs this one record and no dependent fetches.
        int fast = 0;
        rt_f4 p0 = f4(0, 0, 0, 0), p1 = f4(0, 0, 0, 0), i0 = f4(1, 0, 0, 0), i1 = f4(0, 0, 0, 0);
        if (count[0] + count[1] + count[2] + count[3] == 1) {
            if (count[0] == 1) { fast = 1; p0 = out.spheres[first[0]]; }
            else if (count[3] == 1) { fast = 2; p0 = out.boxes[2 * (size_t)first[3]]; p1 = out.boxes[2 * (size_t)first[3] + 1]; }
        }
        if (inst >= 0) { i0 = out.instances[2 * (size_t)inst]; i1 = out.instances[2 * (size_t)inst + 1]; }
        out.media.push_back(f4(i2f(fast), i2f(inst >= 0 ? 1 : 0), 0, 0));
        out.media.push_back(p0); out.media.push_back(p1); out.media.push_back(i0); out.media.push_back(i1);
        return true;
    }
    void emit_tables() {
        for (const MatNode& m : b.materials) {
            int tex = m.tex; const double* rgb = m.albedo;
            if (tex >= 0 && b.textures[tex].type == RT_TEX_SOLID) { rgb = b.textures[tex].rgb; tex = -1; }  // bake SolidColor
            out.materials.push_back(f4(i2f(m.type), i2f(tex), (float)m.fuzz, (float)m.ir));
            // .w = 1 when evaluating the texture is costly (a Noise anywhere in its tree): the shade kernel postpones
            // such hits and evaluates them together
            bool costly = false;
            if (tex >= 0 && (m.type == RT_MAT_LAMBERTIAN || m.type == RT_MAT_LIGHT || m.type == RT_MAT_ISOTROPIC)) {
                std::vector<int> todo(1, tex);
                for (size_t k = 0; k < todo.size() && k < 64; ++k) {
                    const TexNode& t = b.textures[todo[k]];
                    if (t.type == RT_TEX_NOISE) costly = true;
                    if (t.type == RT_TEX_CHECKER) { todo.push_back(t.even); todo.push_back(t.odd); }
                }
            }
            out.materials.push_back(f4((float)rgb[0], (float)rgb[1], (float)rgb[2], i2f(costly ? 1 : 0)));
        }
        for (const TexNode& t : b.textures) {
            if (t.type == RT_TEX_CHECKER) { out.textures.push_back(f4(i2f(t.type), i2f(t.even), i2f(t.odd), 0)); out.textures.push_back(f4(0, 0, 0, 0)); }
            else if (t.type == RT_TEX_NOISE) { out.textures.push_back(f4(i2f(t.type), i2f(t.perlin), 0, 0)); out.textures.push_back(f4((float)t.scale, 0, 0, 0)); }
            else { out.textures.push_back(f4(i2f(t.type), 0, 0, 0)); out.textures.push_back(f4((float)t.rgb[0], (float)t.rgb[1], (float)t.rgb[2], 0)); }
        }
        for (const PerlinTable& p : b.perlins) {
            for (int i = 0; i < 256; ++i) out.perlin_vec.push_back(f4((float)p.ranvec[i][0], (float)p.ranvec[i][1], (float)p.ranvec[i][2], 0));
            for (int k = 0; k < 3; ++k) for (int i = 0; i < 256; ++i) out.perlin_perm.push_back((uint8_t)p.perm[k][i]);
        }
    }

    // Renumbers the inner nodes breadth-first from the top-level root (then from each instance's root), so that
    // "the first K nodes" are the top levels of the tree: the kernels keep that prefix in LDS.
    void relabel_breadth_first() {
        const int n = (int)out.nodes.size() / 4;
        if (n == 0) return;
        std::vector<int> seeds;
        if (out.root >= 0) seeds.push_back(out.root);
        for (size_t i = 0; i < out.instances.size() / RT_INSTANCE_F4; ++i) {
            int32_t r; std::memcpy(&r, &out.instances[i * RT_INSTANCE_F4 + 1].y, 4);
            if (r >= 0) seeds.push_back(r);
        }
        std::vector<int> new_of(n, -1), order; order.reserve(n);
        for (int seed : seeds) {
            if (new_of[seed] >= 0) continue;
            size_t head = order.size();
            new_of[seed] = (int)order.size(); order.push_back(seed);
            while (head < order.size()) {
                int node = order[head++];
                for (int c = 0; c < 2; ++c) {
                    int32_t child; std::memcpy(&child, c == 0 ? &out.nodes[(size_t)node * 4 + 3].x : &out.nodes[(size_t)node * 4 + 3].y, 4);
                    if (child >= 0 && new_of[child] < 0) { new_of[child] = (int)order.size(); order.push_back(child); }
                }
            }
        }
        for (int i = 0; i < n; ++i) if (new_of[i] < 0) { new_of[i] = (int)order.size(); order.push_back(i); }  // unreachable nodes keep a slot
        std::vector<rt_f4> nodes(out.nodes.size());
        for (int i = 0; i < n; ++i) {
            int old = order[i];
            for (int k = 0; k < 4; ++k) nodes[(size_t)i * 4 + k] = out.nodes[(size_t)old * 4 + k];
            int32_t cl, cr; std::memcpy(&cl, &nodes[(size_t)i * 4 + 3].x, 4); std::memcpy(&cr, &nodes[(size_t)i * 4 + 3].y, 4);
            if (cl >= 0) cl = new_of[cl];
            if (cr >= 0) cr = new_of[cr];
            nodes[(size_t)i * 4 + 3].x = i2f(cl); nodes[(size_t)i * 4 + 3].y = i2f(cr);
        }
        out.nodes.swap(nodes);
        if (out.root >= 0) out.root = new_of[out.root];
        for (size_t i = 0; i < out.instances.size() / RT_INSTANCE_F4; ++i) {
            int32_t r; std::memcpy(&r, &out.instances[i * RT_INSTANCE_F4 + 1].y, 4);
            if (r >= 0) out.instances[i * RT_INSTANCE_F4 + 1].y = i2f(new_of[r]);
        }
    }

    // ---- BVH build
    int new_node() { for (int i = 0; i < 4; ++i) out.nodes.push_back(f4(0, 0, 0, 0)); return (int)out.nodes.size() / 4 - 1; }
    void set_node(int n, const Box3& l, const Box3& r, int cl, int cr) {
        rt_f4* q = &out.nodes[(size_t)n * 4];
        q[0] = f4(f_down(l.mn[0]), f_down(l.mn[1]), f_down(l.mn[2]), f_up(l.mx[0]));
        q[1] = f4(f_up(l.mx[1]), f_up(l.mx[2]), f_down(r.mn[0]), f_down(r.mn[1]));
        q[2] = f4(f_down(r.mn[2]), f_up(r.mx[0]), f_up(r.mx[1]), f_up(r.mx[2]));
        q[3] = f4(i2f(cl), i2f(cr), 0, 0);
    }
    Box3 node_box(int n) const {
        const rt_f4* q = &out.nodes[(size_t)n * 4];
        Box3 bx; bx.grow_pt(q[0].x, q[0].y, q[0].z); bx.grow_pt(q[0].w, q[1].x, q[1].y);
        bx.grow_pt(q[1].z, q[1].w, q[2].x); bx.grow_pt(q[2].y, q[2].z, q[2].w);
        return bx;
    }
    int make_leaf(std::vector<Item>& items, size_t lo, size_t hi) {
        if (items[lo].cls == kInstanceClass) return rt_leaf_instance(items[lo].ref);
        int first = -1;
        for (size_t i = lo; i < hi; ++i) { int idx = emit_prim(items[i].ref); if (first < 0) first = idx; }
        return rt_leaf_prims(items[lo].cls, first, (int)(hi - lo));
    }
    static bool leafable(const std::vector<Item>& items, size_t lo, size_t hi) {
        if (hi - lo > RT_LEAF_MAX_PRIMS) return false;
        for (size_t i = lo + 1; i < hi; ++i) if (items[i].cls != items[lo].cls) return false;
        return items[lo].cls != kInstanceClass || hi - lo == 1;
    }
    // Returns a child code for items[lo,hi): a leaf code (<0) or a node index.  `depth` = inner nodes
    // above this subtree; max_depth tracks the deepest inner-node path.
    int build(std::vector<Item>& items, size_t lo, size_t hi, int depth, int budget, int& max_depth, Box3& box_out) {
        Box3 bx, cb;
        for (size_t i = lo; i < hi; ++i) { bx.grow(items[i].box); cb.grow_pt(items[i].cen[0], items[i].cen[1], items[i].cen[2]); }
        box_out = bx;
        size_t n = hi - lo;
        bool can_leaf = leafable(items, lo, hi);
        if (n == 1) return make_leaf(items, lo, hi);

        size_t mid = lo;
        bool split_found = false;
        bool force_median = depth + ceil_log2(n) + 1 >= budget;
        if (!force_median) {
            const int NB = 16;
            double best = kInf; int best_axis = -1, best_bin = -1;
            for (int a = 0; a < 3; ++a) {
                double ext = cb.mx[a] - cb.mn[a];
                if (!(ext > 0)) continue;
                Box3 bb[NB]; size_t cnt[NB] = {0};
                for (size_t i = lo; i < hi; ++i) {
                    int bi = std::min(NB - 1, (int)(NB * (items[i].cen[a] - cb.mn[a]) / ext));
                    bb[bi].grow(items[i].box); cnt[bi]++;
                }
                double ra[NB]; size_t rc[NB]; Box3 acc; size_t c = 0;
                for (int i = NB - 1; i > 0; --i) { acc.grow(bb[i]); c += cnt[i]; ra[i] = acc.area(); rc[i] = c; }
                Box3 l; size_t lc = 0;
                for (int i = 0; i < NB - 1; ++i) {
                    l.grow(bb[i]); lc += cnt[i];
                    if (lc == 0 || rc[i + 1] == 0) continue;
                    double cost = l.area() * (double)lc + ra[i + 1] * (double)rc[i + 1];
                    if (cost < best) { best = cost; best_axis = a; best_bin = i; }
                }
            }
            if (best_axis >= 0) {
                // cost of testing one primitive of the leaf's kind, in node steps of k_trace (measured on final_scene.yml
                // and cornell.yml: the f64 sphere test and the box test weigh about two node steps, a rectangle one)
                const double prim = items[lo].cls == RT_KIND_RECT ? 1.0 : 2.0;
                double leaf_cost = prim * (double)n * bx.area(), split_cost = bx.area() + prim * best;
                if (can_leaf && leaf_cost <= split_cost) return make_leaf(items, lo, hi);
                double ext = cb.mx[best_axis] - cb.mn[best_axis], mnv = cb.mn[best_axis];
                auto it = std::partition(items.begin() + lo, items.begin() + hi, [&](const Item& x) {
                    return std::min(NB - 1, (int)(NB * (x.cen[best_axis] - mnv) / ext)) <= best_bin; });
                mid = (size_t)(it - items.begin());
                split_found = mid > lo && mid < hi;
            }
        }
        if (!split_found) {
            if (can_leaf) return make_leaf(items, lo, hi);
            int axis = 0; double e = -1;
            for (int a = 0; a < 3; ++a) if (cb.mx[a] - cb.mn[a] > e) { e = cb.mx[a] - cb.mn[a]; axis = a; }
            mid = lo + n / 2;
            std::nth_element(items.begin() + lo, items.begin() + mid, items.begin() + hi,
                             [axis](const Item& x, const Item& y) { return x.cen[axis] < y.cen[axis]; });
        }
        int node = new_node();
        max_depth = std::max(max_depth, depth + 1);
        Box3 lb, rb;
        int cl = build(items, lo, mid, depth + 1, budget, max_depth, lb);
        int cr = build(items, mid, hi, depth + 1, budget, max_depth, rb);
        set_node(node, lb, rb, cl, cr);
        return node;
    }
    // Builds a BVH whose root is always an inner node (a lone leaf is referenced by both children).
    int build_root(std::vector<Item>& items, int budget, int& depth) {
        Box3 bx; depth = 0;
        int code = build(items, 0, items.size(), 0, budget, depth, bx);
        if (code >= 0) return code;
        int node = new_node();
        set_node(node, bx, bx, code, code);
        depth = 1;
        return node;
    }
};

}  // namespace

RtSceneView FlatScene::view() const {
    RtSceneView v; std::memset(&v, 0, sizeof(v));
    v.nodes = nodes.data(); v.spheres = spheres.data(); v.sphere_meta = sphere_meta.data();
    v.mspheres = mspheres.data(); v.msphere_meta = msphere_meta.data(); v.rects = rects.data(); v.rect_meta = rect_meta.data();
    v.instances = instances.data(); v.ops = ops.data(); v.media = media.data(); v.materials = materials.data();
    v.textures = textures.data(); v.perlin_vec = perlin_vec.data(); v.perlin_perm = perlin_perm.data(); v.boxes = boxes.data();
    v.root = root;
    v.n_nodes = (int)nodes.size() / 4; v.n_spheres = (int)spheres.size(); v.n_mspheres = (int)msphere_meta.size();
    v.n_rects = (int)rect_meta.size(); v.n_instances = (int)instances.size() / RT_INSTANCE_F4; v.n_ops = (int)ops.size();
    v.n_media = (int)media.size() / RT_MEDIUM_F4; v.n_materials = (int)materials.size() / RT_MATERIAL_F4;
    v.n_textures = (int)textures.size() / RT_TEXTURE_F4; v.n_perlin = (int)perlin_vec.size() / RT_PERLIN_VEC_F4;
    v.max_stack = max_stack; v.n_boxes = (int)boxes.size() / 2;
    v.wnodes = wnodes.data(); v.winst_roots = winst_roots.data(); v.n_wnodes = (int)wnodes.size() / 4; v.wroot = wroot; v.wmax_stack = wmax_stack;
    return v;
}

bool flatten(const Builder& b, HittableObj world, FlatScene& out, std::string& err, bool defer_top) {
    err.clear();
    Flattener f(b, out, err);
    f.defer_top = defer_top;
    return f.run(world);
}

}  // namespace rt
