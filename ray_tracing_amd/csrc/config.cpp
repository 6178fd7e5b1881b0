// config.cpp -- reader for the YAML subset of the reference's config files (configs/*.yml) and the
// mapping onto Config (main.rs:453-494).  Supported YAML: block mappings and block sequences by
// indentation, "key: scalar", "key:" + nested block, "- " items (also at the parent key's indent),
// one-line flow mappings "{x: 1, y: 2}", '#' comments, plain / quoted scalars.  Unknown keys are ignored, as serde does by default.
#include <cstdlib>
#include <fstream>
#include <map>
#include <memory>
#include <sstream>
#include "scene_builder.h"

namespace rt {
namespace {

struct YNode {
    enum Kind { Null, Scalar, Map, Seq } kind = Null;
    std::string scalar;
    std::vector<std::pair<std::string, std::shared_ptr<YNode>>> map;
    std::vector<std::shared_ptr<YNode>> seq;
    const YNode* get(const std::string& k) const {
        if (kind != Map) return nullptr;
        for (auto& kv : map) if (kv.first == k) return kv.second.get();
        return nullptr;
    }
};
struct Line { int indent; std::string text; };

std::string trim(const std::string& s) {
    size_t a = s.find_first_not_of(" \t\r"), b = s.find_last_not_of(" \t\r");
    return a == std::string::npos ? std::string() : s.substr(a, b - a + 1);
}
std::string unquote(const std::string& s) {
    if (s.size() >= 2 && ((s.front() == '"' && s.back() == '"') || (s.front() == '\'' && s.back() == '\''))) return s.substr(1, s.size() - 2);
    return s;
}

class Parser {
public:
    std::vector<Line> lines; size_t pos = 0; std::string err;
    explicit Parser(const std::string& text) {
        std::istringstream in(text); std::string raw;
        while (std::getline(in, raw)) {
            std::string s; bool q1 = false, q2 = false;
            for (size_t i = 0; i < raw.size(); ++i) {  // strip comments outside quotes
                char c = raw[i];
                if (c == '\'' && !q2) q1 = !q1; else if (c == '"' && !q1) q2 = !q2;
                if (c == '#' && !q1 && !q2 && (i == 0 || raw[i - 1] == ' ' || raw[i - 1] == '\t')) break;
                s.push_back(c);
            }
            if (trim(s).empty() || trim(s) == "---") continue;
            int ind = 0; while (ind < (int)s.size() && s[ind] == ' ') ++ind;
            if (ind < (int)s.size() && s[ind] == '\t') { err = "tabs are not allowed for indentation"; return; }
            Line l = {ind, trim(s)}; lines.push_back(l);
        }
    }
    static bool is_dash(const std::string& t) { return t == "-" || (t.size() > 1 && t[0] == '-' && t[1] == ' '); }
    std::shared_ptr<YNode> parse_node(int indent) {
        if (pos >= lines.size()) return std::make_shared<YNode>();
        return is_dash(lines[pos].text) ? parse_seq(indent) : parse_map(indent);
    }
    std::shared_ptr<YNode> parse_map(int indent) {
        auto n = std::make_shared<YNode>(); n->kind = YNode::Map;
        while (pos < lines.size() && lines[pos].indent == indent && !is_dash(lines[pos].text) && err.empty()) {
            const std::string& t = lines[pos].text;
            size_t c = t.find(':');
            if (c == std::string::npos || (c + 1 < t.size() && t[c + 1] != ' ')) { err = "expected 'key: value' at '" + t + "'"; return n; }
            std::string key = unquote(trim(t.substr(0, c))), val = trim(t.substr(c + 1));
            ++pos;
            std::shared_ptr<YNode> child;
            if (!val.empty() && val[0] == '{') child = parse_flow_map(val);
            else if (!val.empty()) { child = std::make_shared<YNode>(); child->kind = YNode::Scalar; child->scalar = unquote(val); }
            else if (pos < lines.size() && lines[pos].indent > indent) child = parse_node(lines[pos].indent);
            else if (pos < lines.size() && lines[pos].indent == indent && is_dash(lines[pos].text)) child = parse_seq(indent);
            else child = std::make_shared<YNode>();
            n->map.push_back(std::make_pair(key, child));
        }
        if (pos < lines.size() && lines[pos].indent > indent && err.empty()) err = "bad indentation at '" + lines[pos].text + "'";
        return n;
    }
    // Flow mapping "{k: v, k2: {..}}" on one line.
    std::shared_ptr<YNode> parse_flow_map(const std::string& text) {
        auto n = std::make_shared<YNode>(); n->kind = YNode::Map;
        std::string t = trim(text);
        if (t.size() < 2 || t.front() != '{' || t.back() != '}') { err = "bad flow mapping '" + text + "'"; return n; }
        std::string body = t.substr(1, t.size() - 2);
        std::vector<std::string> parts; int depth = 0; std::string cur;
        for (char c : body) {
            if (c == '{') ++depth; else if (c == '}') --depth;
            if (c == ',' && depth == 0) { parts.push_back(cur); cur.clear(); } else cur.push_back(c);
        }
        if (!trim(cur).empty()) parts.push_back(cur);
        for (const std::string& part : parts) {
            size_t c = part.find(':');
            if (c == std::string::npos) { err = "bad flow mapping entry '" + part + "'"; return n; }
            std::string key = unquote(trim(part.substr(0, c))), val = trim(part.substr(c + 1));
            std::shared_ptr<YNode> child;
            if (!val.empty() && val[0] == '{') child = parse_flow_map(val);
            else { child = std::make_shared<YNode>(); child->kind = val.empty() ? YNode::Null : YNode::Scalar; child->scalar = unquote(val); }
            n->map.push_back(std::make_pair(key, child));
        }
        return n;
    }
    std::shared_ptr<YNode> parse_seq(int indent) {
        auto n = std::make_shared<YNode>(); n->kind = YNode::Seq;
        while (pos < lines.size() && lines[pos].indent == indent && is_dash(lines[pos].text) && err.empty()) {
            std::string rest = lines[pos].text.size() > 1 ? trim(lines[pos].text.substr(2)) : std::string();
            if (rest.empty()) {
                ++pos;
                if (pos < lines.size() && lines[pos].indent > indent) n->seq.push_back(parse_node(lines[pos].indent));
                else n->seq.push_back(std::make_shared<YNode>());
            } else if (rest[0] == '{') {
                n->seq.push_back(parse_flow_map(rest)); ++pos;
            } else if (rest.find(": ") != std::string::npos || rest.back() == ':') {
                // "- key: value": the item is a mapping whose entries sit at the column after "- "
                int inner = indent + 2 + (int)(lines[pos].text.substr(2).find_first_not_of(' '));
                lines[pos].indent = inner; lines[pos].text = rest;
                n->seq.push_back(parse_map(inner));
            } else {
                auto s = std::make_shared<YNode>(); s->kind = YNode::Scalar; s->scalar = unquote(rest);
                n->seq.push_back(s); ++pos;
            }
        }
        return n;
    }
};

bool get_f64(const YNode* n, double& out) {
    if (!n || n->kind != YNode::Scalar) return false;
    const char* s = n->scalar.c_str(); char* end = nullptr;
    double v = std::strtod(s, &end);
    if (end == s || *end != '\0') return false;
    out = v; return true;
}
bool get_int(const YNode* n, long long& out) {  // serde: usize / u32 fields reject negatives and fractions
    if (!n || n->kind != YNode::Scalar) return false;
    const char* s = n->scalar.c_str(); char* end = nullptr;
    long long v = std::strtoll(s, &end, 10);
    if (end == s || *end != '\0' || v < 0) return false;
    out = v; return true;
}
bool get_vec(const YNode* n, Vec3& out) {  // utils.rs:92-97 SerdeVector {x, y, z}
    if (!n || n->kind != YNode::Map) return false;
    return get_f64(n->get("x"), out.x) && get_f64(n->get("y"), out.y) && get_f64(n->get("z"), out.z);
}
// ---- "List" scenes beyond the reference's schema.  The reference's List holds spheres with three materials
// (objects.rs:102-107, materials.rs:34-40; "TODO make this more generic", objects.rs:102).  Here an item may carry a `type`:
//   Sphere (default) {center, radius, material}        MovingSphere {center0, center1, time0, time1, radius, material}
//   Rectangle {plane: Xy|Xz|Yz, x0, x1, y0, y1, k, material}        Box {min, max, material}
//   Translate {offset, object}    RotateY {angle_deg, object}    ConstantMedium {boundary, density, albedo}    List {objects}
// material: Lambertian {albedo | texture}, Metal {albedo, fuzz}, Dielectric {ir}, DiffuseLight {color | texture}, Isotropic {albedo | texture}
// texture:  a colour {x, y, z}, or {type: SolidColor, color} | {type: Checker, even, odd} | {type: Noise, scale}
// A Noise draws its Perlin tables from the scene stream, in document order.
struct SceneReader {
    Builder& b; SceneRng& rng; std::string err;
    SceneReader(Builder& b_, SceneRng& r) : b(b_), rng(r) {}
    int fail(const std::string& m) { if (err.empty()) err = m; return -1; }
    static std::string type_of(const YNode* n, const char* dflt) {
        const YNode* t = n ? n->get("type") : nullptr;
        return t && t->kind == YNode::Scalar ? t->scalar : std::string(dflt);
    }
    int texture(const YNode* n) {
        if (!n || n->kind != YNode::Map) return fail("texture: expected a mapping");
        Vec3 c;
        if (!n->get("type")) { if (!get_vec(n, c)) return fail("texture: expected {x, y, z} or a typed texture"); return b.solid_color(c); }
        std::string t = type_of(n, "");
        if (t == "SolidColor") { if (!get_vec(n->get("color"), c)) return fail("SolidColor: color"); return b.solid_color(c); }
        if (t == "Checker") {
            int e = texture(n->get("even")); if (e < 0) return -1;
            int o = texture(n->get("odd")); if (o < 0) return -1;
            return b.checker(e, o);
        }
        if (t == "Noise") { double sc; if (!get_f64(n->get("scale"), sc)) return fail("Noise: scale"); return b.noise(sc, rng); }
        return fail("unknown texture type '" + t + "'");
    }
    int tex_or_color(const YNode* m, const char* color_key) {      // {texture: ...} wins over {<color_key>: {x, y, z}}
        if (m->get("texture")) return texture(m->get("texture"));
        Vec3 c;
        if (!get_vec(m->get(color_key), c)) return fail(std::string("material: ") + color_key);
        return b.solid_color(c);
    }
    int material(const YNode* m) {
        if (!m || m->kind != YNode::Map) return fail("material: expected a mapping");
        std::string t = type_of(m, "");
        Vec3 c; double f;
        if (t == "Lambertian") { int x = tex_or_color(m, "albedo"); return x < 0 ? -1 : b.lambertian_from_texture(x); }
        if (t == "Metal") { if (!get_vec(m->get("albedo"), c) || !get_f64(m->get("fuzz"), f)) return fail("Metal: albedo, fuzz"); return b.metal(c, f); }
        if (t == "Dielectric") { if (!get_f64(m->get("ir"), f)) return fail("Dielectric: ir"); return b.dielectric(f); }
        if (t == "DiffuseLight") { int x = tex_or_color(m, "color"); return x < 0 ? -1 : b.diffuse_light(x); }
        if (t == "Isotropic") { int x = tex_or_color(m, "albedo"); return x < 0 ? -1 : b.isotropic(x); }
        return fail("unknown material type '" + t + "'");
    }
    int list(const YNode* seq) {
        if (!seq || seq->kind != YNode::Seq) return fail("objects: expected a sequence");
        std::vector<HittableObj> kids;
        for (auto& item : seq->seq) { int o = object(item.get()); if (o < 0) return -1; kids.push_back(o); }
        return b.list(kids);
    }
    int object(const YNode* n) {
        if (!n || n->kind != YNode::Map) return fail("object: expected a mapping");
        std::string t = type_of(n, "Sphere");
        Vec3 a, c; double r, t0, t1;
        if (t == "Sphere") {
            if (!get_vec(n->get("center"), a) || !get_f64(n->get("radius"), r)) return fail("Sphere: center, radius");
            int m = material(n->get("material")); return m < 0 ? -1 : b.sphere(a, r, m);
        }
        if (t == "MovingSphere") {
            if (!get_vec(n->get("center0"), a) || !get_vec(n->get("center1"), c) || !get_f64(n->get("time0"), t0) ||
                !get_f64(n->get("time1"), t1) || !get_f64(n->get("radius"), r)) return fail("MovingSphere: center0, center1, time0, time1, radius");
            int m = material(n->get("material")); return m < 0 ? -1 : b.moving_sphere(a, c, t0, t1, r, m);
        }
        if (t == "Rectangle") {
            double x0, x1, y0, y1, k;
            const YNode* pl = n->get("plane");
            if (!pl || pl->kind != YNode::Scalar || !get_f64(n->get("x0"), x0) || !get_f64(n->get("x1"), x1) || !get_f64(n->get("y0"), y0) ||
                !get_f64(n->get("y1"), y1) || !get_f64(n->get("k"), k)) return fail("Rectangle: plane, x0, x1, y0, y1, k");
            RectangleType rt_ = pl->scalar == "Xy" ? Xy : pl->scalar == "Xz" ? Xz : Yz;
            if (pl->scalar != "Xy" && pl->scalar != "Xz" && pl->scalar != "Yz") return fail("Rectangle: plane must be Xy, Xz or Yz");
            int m = material(n->get("material")); return m < 0 ? -1 : b.rectangle(m, x0, x1, y0, y1, k, rt_);
        }
        if (t == "Box") {
            if (!get_vec(n->get("min"), a) || !get_vec(n->get("max"), c)) return fail("Box: min, max");
            int m = material(n->get("material")); return m < 0 ? -1 : b.box(a, c, m);
        }
        if (t == "Translate") {
            if (!get_vec(n->get("offset"), a)) return fail("Translate: offset");
            int o = object(n->get("object")); return o < 0 ? -1 : b.translate(o, a);
        }
        if (t == "RotateY") {
            if (!get_f64(n->get("angle_deg"), r)) return fail("RotateY: angle_deg");
            int o = object(n->get("object")); return o < 0 ? -1 : b.rotate_y(o, r);
        }
        if (t == "ConstantMedium") {
            if (!get_f64(n->get("density"), r)) return fail("ConstantMedium: density");
            int o = object(n->get("boundary")); if (o < 0) return -1;
            int x = texture(n->get("albedo")); return x < 0 ? -1 : b.constant_medium(o, r, x);
        }
        if (t == "List") return list(n->get("objects"));
        return fail("unknown object type '" + t + "'");
    }
};

}  // namespace

HittableObj list_scene_from_yaml(Builder& b, const std::string& text, SceneRng& rng) {
    Parser p(text);
    std::shared_ptr<YNode> root;
    if (p.err.empty()) root = p.parse_node(p.lines.empty() ? 0 : p.lines[0].indent);
    if (!p.err.empty() || !root || root->kind != YNode::Map) { b.error = "Could not read values."; return -1; }
    const YNode* scene = root->get("scene");
    const YNode* objs = scene ? scene->get("objects") : (root->get("objects") ? root->get("objects")->get("objects") : nullptr);
    SceneReader rd(b, rng);
    int w = rd.list(objs);
    if (w < 0) b.error = "Could not read values. (scene.objects: " + (rd.err.empty() ? b.error : rd.err) + ")";
    return w;
}

// A List scene is built once into a scratch builder so that a bad object is reported when the config is read.
static bool check_list(const std::string& text, Config& cfg, std::string& err) {
    Builder scratch; SceneRng rng(0);
    int w = list_scene_from_yaml(scratch, text, rng);
    if (w < 0) { err = scratch.error; return false; }
    cfg.n_list_objects = (int)scratch.objects[w].children.size();
    return true;
}

bool parse_config(const std::string& text, Config& cfg, std::string& err) {
    Parser p(text);
    std::shared_ptr<YNode> root;
    if (p.err.empty()) root = p.parse_node(p.lines.empty() ? 0 : p.lines[0].indent);
    if (!p.err.empty() || !root || root->kind != YNode::Map || p.pos != p.lines.size()) {
        err = "Could not read values." + (p.err.empty() ? std::string() : " (" + p.err + ")"); return false;  // main.rs:512
    }
    cfg = Config();
    cfg.source = text;
    long long iv;
    if (root->get("image_width")) { if (!get_int(root->get("image_width"), iv)) { err = "Could not read values. (image_width)"; return false; } cfg.image_width = (int)iv; }
    if (root->get("samples_per_pixel")) { if (!get_int(root->get("samples_per_pixel"), iv)) { err = "Could not read values. (samples_per_pixel)"; return false; } cfg.samples_per_pixel = (int)iv; }
    if (root->get("max_depth")) { if (!get_int(root->get("max_depth"), iv)) { err = "Could not read values. (max_depth)"; return false; } cfg.max_depth = (int)iv; }
    const YNode* cam = root->get("camera");  // cameras.rs:10-18: all seven fields are required
    if (!cam || !get_vec(cam->get("look_from"), cfg.camera.look_from) || !get_vec(cam->get("look_at"), cfg.camera.look_at) ||
        !get_vec(cam->get("v_up"), cfg.camera.v_up) || !get_f64(cam->get("vertical_fov_deg"), cfg.camera.vertical_fov_deg) ||
        !get_f64(cam->get("aspect_ratio"), cfg.camera.aspect_ratio) || !get_f64(cam->get("aperture"), cfg.camera.aperture) ||
        !get_f64(cam->get("focus_distance"), cfg.camera.focus_distance)) {
        err = "Could not read values. (camera)"; return false;
    }
    // Not in the reference's schema (its TODOs at main.rs:533-534, 544): background colour and shutter interval.
    Vec3 bg;
    if (root->get("background")) { if (!get_vec(root->get("background"), bg)) { err = "Could not read values. (background)"; return false; } cfg.background = bg; }
    if (cam->get("shutter_open") && !get_f64(cam->get("shutter_open"), cfg.shutter_open)) { err = "Could not read values. (camera.shutter_open)"; return false; }
    if (cam->get("shutter_close") && !get_f64(cam->get("shutter_close"), cfg.shutter_close)) { err = "Could not read values. (camera.shutter_close)"; return false; }
    const YNode* scene = root->get("scene");  // main.rs:483-494, internally tagged by "type"
    if (scene) {
        const YNode* ty = scene->get("type");
        if (!ty || ty->kind != YNode::Scalar) { err = "Could not read values. (scene.type)"; return false; }
        cfg.scene_type = ty->scalar;
        static const char* known[] = {"Random", "TwoSpheres", "TwoPerlinSpheres", "SimpleLight", "Cornell", "CornellSmoke", "FinalScene", "List"};
        bool ok = false; for (const char* k : known) ok = ok || cfg.scene_type == k;
        if (!ok) { err = "Could not read values. (unknown scene type '" + cfg.scene_type + "')"; return false; }
        if (cfg.scene_type == "List" && !check_list(text, cfg, err)) return false;
    } else if (const YNode* legacy = root->get("objects")) {
        // Older schema of configs/orig_img.yml: top-level `objects: {objects: [...]}` and no `scene:`.
        // The reference at HEAD rejects it (missing field `scene`); it is accepted here as a List scene.
        cfg.scene_type = "List";
        (void)legacy;
        if (!check_list(text, cfg, err)) return false;
    } else { err = "Could not read values. (missing field `scene`)"; return false; }
    return true;
}

bool load_config(const std::string& path, Config& cfg, std::string& err) {
    std::ifstream f(path.c_str());
    if (!f) { err = "Could not open config file."; return false; }  // main.rs:511
    std::stringstream ss; ss << f.rdbuf();
    return parse_config(ss.str(), cfg, err);
}

}  // namespace rt
