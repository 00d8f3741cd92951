// rt_host.cpp — scene description, OBJ loading, mesh transforms, scene catalogue and flattening.
// Citations are paths under the reference root (raytracer-rs).

#include "rt_host.hpp"
#include "../../include/rt_host.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>
#include <string>
#include <vector>

namespace rt {

static const float PI_F = 3.14159265358979323846f;

float dot(Vec3 a, Vec3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
Vec3 cross(Vec3 a, Vec3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
Vec3 normalize(Vec3 a) { return a * (1.0f / std::sqrt(dot(a, a))); }

SamplerKind sampler_kind_from_int(int i) {
    switch (i) { // sampler.rs:13-21
    case 1: return SamplerKind::WhiteNoise;
    case 2: return SamplerKind::Uniform;
    case 3: return SamplerKind::Jittered;
    case 4: return SamplerKind::BlueNoise;
    default: throw std::invalid_argument("unexpected integer");
    }
}

// ---------------------------------------------------------------------------------------------------- materials
static rt_material blank_material(int kind) {
    rt_material m;
    std::memset(&m, 0, sizeof(m));
    m.kind = kind;
    return m;
}
rt_material Diffuse::describe() const {
    rt_material m = blank_material(RT_MAT_DIFFUSE);
    m.albedo[0] = albedo.x; m.albedo[1] = albedo.y; m.albedo[2] = albedo.z;
    return m;
}
rt_material Metal::describe() const {
    rt_material m = blank_material(RT_MAT_METAL);
    m.albedo[0] = albedo.x; m.albedo[1] = albedo.y; m.albedo[2] = albedo.z;
    return m;
}
rt_material Dielectric::describe() const {
    rt_material m = blank_material(RT_MAT_DIELECTRIC);
    m.ref_idx = ref_idx;
    return m;
}
rt_material Microfacet::describe() const {
    rt_material m = blank_material(RT_MAT_MICROFACET);
    m.albedo[0] = attenuation.x; m.albedo[1] = attenuation.y; m.albedo[2] = attenuation.z;
    m.f0[0] = f0.x; m.f0[1] = f0.y; m.f0[2] = f0.z;
    m.roughness = roughness;
    m.metallic = metallic;
    return m;
}

// ---------------------------------------------------------------------------------------------------- flattening
int Flattener::material_index(const MaterialRef &m) {
    if (!m) throw std::invalid_argument("no material"); // shader.rs:46,81,106
    rt_material d = m->describe();
    for (size_t i = 0; i < out.mats.size(); i++)
        if (std::memcmp(&out.mats[i], &d, sizeof(d)) == 0) return (int)i;
    out.mats.push_back(d);
    return (int)out.mats.size() - 1;
}

static void push_tri(std::vector<float> &v, const Triangle &t) {
    for (int k = 0; k < 3; k++) {
        v.push_back(t.vertex[k].x);
        v.push_back(t.vertex[k].y);
        v.push_back(t.vertex[k].z);
    }
}

Vec3 Triangle::normal() const { return normalize(cross(vertex[1] - vertex[0], vertex[2] - vertex[0])); }
float Triangle::area() const {
    Vec3 c = cross(vertex[1] - vertex[0], vertex[2] - vertex[0]);
    return 0.5f * std::sqrt(dot(c, c));
}

void Triangle::flatten(Flattener &f) const {
    int first = (int)f.out.tri_mat.size();
    push_tri(f.out.tri_v, *this);
    f.out.tri_mat.push_back(f.material_index(mat));
    f.out.objects.push_back(rt_object{RT_OBJ_TRIANGLE, first, 1, 0});
}
void Sphere::flatten(Flattener &f) const {
    int first = (int)f.out.sph_mat.size();
    f.out.sph.insert(f.out.sph.end(), {center.x, center.y, center.z, radius});
    f.out.sph_mat.push_back(f.material_index(mat));
    f.out.objects.push_back(rt_object{RT_OBJ_SPHERE, first, 1, 0});
}
void Cylinder::flatten(Flattener &f) const {
    int first = (int)f.out.cyl_mat.size();
    f.out.cyl.insert(f.out.cyl.end(), {center_x, center_z, y_min, y_max, radius});
    f.out.cyl_mat.push_back(f.material_index(mat));
    f.out.objects.push_back(rt_object{RT_OBJ_CYLINDER, first, 1, 0});
}
void MeshBase::flatten(Flattener &f) const {
    int first = (int)f.out.tri_mat.size();
    int last_idx = -1;
    const Material *last = nullptr;
    for (const Triangle &t : face_list) {
        push_tri(f.out.tri_v, t);
        if (t.mat.get() != last || last_idx < 0) {
            last = t.mat.get();
            last_idx = f.material_index(t.mat);
        }
        f.out.tri_mat.push_back(last_idx);
    }
    f.out.objects.push_back(rt_object{RT_OBJ_MESH, first, (int)face_list.size(), accel_kind()});
}

void PointLight::flatten(Flattener &f) const {
    rt_light l;
    std::memset(&l, 0, sizeof(l));
    l.kind = RT_LIGHT_POINT;
    l.origin[0] = origin.x; l.origin[1] = origin.y; l.origin[2] = origin.z;
    l.spectrum[0] = spectrum.x; l.spectrum[1] = spectrum.y; l.spectrum[2] = spectrum.z;
    l.sampler_kind = RT_SAMPLER_WHITE;
    f.out.lights.push_back(l);
}
void DiskLight::flatten(Flattener &f) const {
    rt_light l;
    std::memset(&l, 0, sizeof(l));
    l.kind = RT_LIGHT_DISK;
    l.origin[0] = origin.x; l.origin[1] = origin.y; l.origin[2] = origin.z;
    l.radius = radius;
    l.spectrum[0] = spectrum.x; l.spectrum[1] = spectrum.y; l.spectrum[2] = spectrum.z;
    l.sampler_kind = (int)sampler_kind;
    f.out.lights.push_back(l);
}

static bool vec_eq(Vec3 a, Vec3 b) { // lib.rs:31-34
    Vec3 d = a - b;
    return dot(d, d) < 0.001f;
}

PolygonLight::PolygonLight(NaiveMesh m, RGBSpectrum s) : mesh(std::move(m)), spectrum(s) {
    if (mesh.face_list.empty()) throw std::invalid_argument("PolygonLight needs at least one face");
    Vec3 n = mesh.face_list[0].normal();
    for (const Triangle &t : mesh.face_list)
        if (!vec_eq(n, t.normal())) throw std::invalid_argument("the mesh of PolygonLight is not 2D!"); // light.rs:258-261
}
void PolygonLight::flatten(Flattener &f) const {
    rt_light l;
    std::memset(&l, 0, sizeof(l));
    l.kind = RT_LIGHT_POLYGON;
    l.spectrum[0] = spectrum.x; l.spectrum[1] = spectrum.y; l.spectrum[2] = spectrum.z;
    l.tri_first = (int)(f.out.light_tri_v.size() / 9);
    l.tri_count = (int)mesh.face_list.size();
    l.sampler_kind = RT_SAMPLER_WHITE;
    for (const Triangle &t : mesh.face_list) push_tri(f.out.light_tri_v, t);
    f.out.lights.push_back(l);
}

void FlatScene::finish() {
    desc.n_tri = (int)tri_mat.size();
    desc.tri_v = tri_v.data();
    desc.tri_mat = tri_mat.data();
    desc.n_sph = (int)sph_mat.size();
    desc.sph = sph.data();
    desc.sph_mat = sph_mat.data();
    desc.n_cyl = (int)cyl_mat.size();
    desc.cyl = cyl.data();
    desc.cyl_mat = cyl_mat.data();
    desc.n_mat = (int)mats.size();
    desc.mats = mats.data();
    desc.n_light = (int)lights.size();
    desc.lights = lights.data();
    desc.n_light_tri = (int)(light_tri_v.size() / 9);
    desc.light_tri_v = light_tri_v.data();
    desc.n_obj = (int)objects.size();
    desc.objects = objects.data();
}

std::unique_ptr<FlatScene> Scene::flatten() const {
    std::unique_ptr<FlatScene> fs(new FlatScene());
    Flattener f{*fs, {}};
    for (const auto &o : world.objects.list) o->flatten(f);
    for (const auto &l : world.lights.list) l->flatten(f);
    fs->finish();
    return fs;
}

// ---------------------------------------------------------------------------------------------------- camera
Camera::Camera(Pt3 lookfrom, Pt3 lookat, Vec3 vup, float vfov, float aspect) { // camera.rs:13-27
    float theta = vfov * (PI_F / 180.0f);
    float half_height = std::tan(theta / 2.0f);
    float half_width = aspect * half_height;
    origin = lookfrom;
    Vec3 w = normalize(lookfrom - lookat);
    Vec3 u = normalize(cross(vup, w));
    Vec3 v = cross(w, u);
    lower_left_corner = Pt3(-half_width, -half_height, -1.0f);
    horizontal = 2.0f * half_width * u;
    vertical = 2.0f * half_height * v;
}
rt_camera Camera::describe() const {
    rt_camera c;
    c.origin[0] = origin.x; c.origin[1] = origin.y; c.origin[2] = origin.z;
    c.llc[0] = lower_left_corner.x; c.llc[1] = lower_left_corner.y; c.llc[2] = lower_left_corner.z;
    c.horizontal[0] = horizontal.x; c.horizontal[1] = horizontal.y; c.horizontal[2] = horizontal.z;
    c.vertical[0] = vertical.x; c.vertical[1] = vertical.y; c.vertical[2] = vertical.z;
    return c;
}

// ---------------------------------------------------------------------------------------------------- transforms
// mesh.rs:23-39: Decomposed{scale, rot: Quaternion::from(Euler{Deg x, Deg y, Deg z}), disp}.transform_point(p)
//  = rot.rotate_point(p * scale) + disp, with cgmath 0.18's XYZ quaternion (q = qx * qy * qz).
struct Quat { float s, x, y, z; };
static Quat quat_from_euler_deg(float xd, float yd, float zd) {
    float hx = xd * (PI_F / 180.0f) * 0.5f, hy = yd * (PI_F / 180.0f) * 0.5f, hz = zd * (PI_F / 180.0f) * 0.5f;
    float sx = std::sin(hx), cx = std::cos(hx), sy = std::sin(hy), cy = std::cos(hy), sz = std::sin(hz), cz = std::cos(hz);
    return Quat{-sx * sy * sz + cx * cy * cz, sx * cy * cz + sy * sz * cx, -sx * sz * cy + sy * cx * cz, sx * sy * cz + sz * cx * cy};
}
static Vec3 quat_rotate(const Quat &q, Vec3 v) {
    Vec3 qv(q.x, q.y, q.z);
    Vec3 tmp = cross(qv, v) + v * q.s;
    return cross(qv, tmp) * 2.0f + v;
}
static Vec3 transform_point(const Quat &q, float scale, Vec3 disp, Vec3 p) { return quat_rotate(q, p * scale) + disp; }

void MeshBase::transform(float scale, Vec3 disp, float x, float y, float z) {
    Quat q = quat_from_euler_deg(x, y, z);
    for (Triangle &f : face_list)
        for (int k = 0; k < 3; k++) f.vertex[k] = transform_point(q, scale, disp, f.vertex[k]);
}

// ---------------------------------------------------------------------------------------------------- OBJ loader
// obj_loader.rs:16-47 with obj-rs 0.6.4's load_obj::<Position, usize>: positions in file order, every face must be a
// triangle, indices are 1-based (negative = relative to the end), "a", "a/b", "a//c", "a/b/c" all take `a`.
static std::vector<float> parse_obj(const std::string &path) {
    std::ifstream in(path);
    if (!in) throw std::runtime_error("cannot open " + path);
    std::vector<Vec3> pos;
    std::vector<float> tri;
    std::string line;
    size_t lineno = 0;
    while (std::getline(in, line)) {
        lineno++;
        std::istringstream ss(line);
        std::string tag;
        if (!(ss >> tag) || tag[0] == '#') continue;
        if (tag == "v") {
            Vec3 p;
            if (!(ss >> p.x >> p.y >> p.z)) throw std::runtime_error(path + ":" + std::to_string(lineno) + ": bad vertex");
            pos.push_back(p);
        } else if (tag == "f") {
            std::vector<long> idx;
            std::string tok;
            while (ss >> tok) {
                long a = std::strtol(tok.c_str(), nullptr, 10);
                if (a == 0) throw std::runtime_error(path + ":" + std::to_string(lineno) + ": bad face index");
                if (a < 0) a = (long)pos.size() + a + 1;
                if (a < 1 || a > (long)pos.size()) throw std::runtime_error(path + ":" + std::to_string(lineno) + ": face index out of range");
                idx.push_back(a - 1);
            }
            if (idx.size() != 3) throw std::runtime_error("Model should be triangulated first to be loaded properly");
            for (long a : idx) {
                tri.push_back(pos[a].x);
                tri.push_back(pos[a].y);
                tri.push_back(pos[a].z);
            }
        }
        // vn, vt, g, o, s, usemtl, mtllib: ignored for Position-only loading
    }
    return tri;
}

FastMesh load_obj_file_ref(const std::string &path, MaterialRef mat) {
    std::vector<float> tri = parse_obj(path);
    std::vector<Triangle> list;
    list.reserve(tri.size() / 9);
    for (size_t i = 0; i + 8 < tri.size(); i += 9)
        list.emplace_back(Pt3(tri[i], tri[i + 1], tri[i + 2]), Pt3(tri[i + 3], tri[i + 4], tri[i + 5]), Pt3(tri[i + 6], tri[i + 7], tri[i + 8]), mat);
    return FastMesh(std::move(list));
}

// ---------------------------------------------------------------------------------------------------- scene helpers
std::pair<Triangle, Triangle> make_square(Pt3 a, Pt3 b, Pt3 c, Pt3 d, RGBSpectrum albedo) { // scene.rs:19-29
    auto m = std::make_shared<Diffuse>(albedo);
    return {Triangle(a, b, c, m), Triangle(a, c, d, m)};
}
std::pair<Triangle, Triangle> make_mf_square(Pt3 a, Pt3 b, Pt3 c, Pt3 d, const Microfacet &mf) { // scene.rs:31-41
    auto m = std::make_shared<Microfacet>(mf);
    return {Triangle(a, b, c, m), Triangle(a, c, d, m)};
}
std::vector<Triangle> make_chess_board(Pt3 v0, Pt3 v1, Pt3 v2, Pt3 v3, int num) { // scene.rs:43-61
    (void)v2;
    Vec3 ori = v0;
    Vec3 vi = (v1 - v0) / (float)num;
    Vec3 vj = (v3 - v0) / (float)num;
    std::vector<Triangle> list;
    for (int i = 1; i < num + 1; i++) {
        for (int j = 1; j < num + 1; j++) {
            Pt3 p00 = ori + (float)(i - 1) * vi + (float)(j - 1) * vj;
            Pt3 p01 = ori + (float)(i - 1) * vi + (float)j * vj;
            Pt3 p10 = ori + (float)i * vi + (float)(j - 1) * vj;
            Pt3 p11 = ori + (float)i * vi + (float)j * vj;
            RGBSpectrum color = ((i + j) % 2 == 0) ? BLACK : WHITE;
            auto sq = make_square(p00, p10, p11, p01, color);
            list.push_back(sq.first);
            list.push_back(sq.second);
        }
    }
    return list;
}

namespace {

// the box of scene.rs:119-135 in world-list order t1..t6, t7..t10 (scene.rs:178-188)
void push_cornell_walls(World &w) {
    Pt3 v1(-1, -1, -1), v2(1, -1, -1), v3(1, -1, -3), v4(-1, -1, -3);
    Pt3 v5(-1, 1, -1), v6(1, 1, -1), v7(1, 1, -3), v8(-1, 1, -3);
    auto a = make_square(v1, v2, v3, v4, RGBSpectrum(0.9f, 0.9f, 0.9f));
    auto b = make_square(v5, v8, v7, v6, RGBSpectrum(0.9f, 0.9f, 0.9f));
    auto c = make_square(v4, v3, v7, v8, RGBSpectrum(0.9f, 0.9f, 0.9f));
    auto d = make_square(v1, v4, v8, v5, RGBSpectrum(0.7f, 0.0f, 0.0f));
    auto e = make_square(v3, v2, v6, v7, RGBSpectrum(0.0f, 0.4f, 0.0f));
    for (auto *p : {&a, &b, &c, &d, &e}) {
        w.objects.push(p->first);
        w.objects.push(p->second);
    }
}

// the ceiling light of scene.rs:144-158,345-348
void push_ceiling_light(World &w) {
    auto sq = make_square(Pt3(-0.22f, 1.0f, -2.22f), Pt3(0.22f, 1.0f, -2.22f), Pt3(0.22f, 1.0f, -1.78f), Pt3(-0.22f, 1.0f, -1.78f), BLACK);
    NaiveMesh m(std::vector<Triangle>{sq.first, sq.second});
    w.lights.push(PolygonLight(std::move(m), RGBSpectrum(0.9f, 0.64f, 0.28f) * 22.0f));
}

Camera cornell_camera(float aspect) { // scene.rs:160-169
    return Camera(Pt3(0.0f, 0.0f, 2.0f), Pt3(0.0f, 0.0f, -1.0f), Vec3(0.0f, 1.0f, 0.0f), 90.0f, aspect);
}

// materials that appear (live or commented) in scene.rs:65-118,189-218
Microfacet mf_pyramid() { return Microfacet(RGBSpectrum(0.98f, 0.98f, 0.98f), 0.6f, 0.5f, RGBSpectrum(0.8f, 0.8f, 0.8f)); }        // :71-76
Microfacet mf_miku() { return Microfacet(RGBSpectrum(0.08f, 0.08f, 0.08f), 0.3f, 0.1f, RGBSpectrum(1.0f, 0.7f, 0.9f)); }           // :93-98
Microfacet mf_cylinder() { return Microfacet(RGBSpectrum(0.98f, 0.98f, 0.98f), 0.13f, 0.9f, RGBSpectrum(0.9f, 0.7f, 0.4f)); }      // :198-203
Microfacet mf_sphere() { return Microfacet(RGBSpectrum(0.09f, 0.09f, 0.59f), 0.05f, 0.1f, RGBSpectrum(0.4f, 0.7f, 0.9f)); }        // :212-217

} // namespace

// scene.rs:64-354 as it stands at HEAD.  The four meshes loaded at :65-118 are never placed in the world
// (:173-177 are comments) and two of them are not in the repository, so they are not loaded here.
Scene Scene::cornell_box(SamplerKind, float aspect) {
    Scene s;
    s.cam = cornell_camera(aspect);
    push_cornell_walls(s.world);
    struct Sp { float x, y; float f0; float rough; float metal; RGBSpectrum att; };
    const Sp sp[9] = {
        {-0.6f, -0.75f, 0.89f, 0.05f, 0.8f, RGBSpectrum(0.9f, 0.7f, 0.9f)}, {0.0f, -0.75f, 0.89f, 0.2f, 0.8f, RGBSpectrum(0.4f, 0.7f, 0.9f)},
        {0.6f, -0.75f, 0.89f, 0.5f, 0.8f, RGBSpectrum(0.4f, 0.9f, 0.9f)},   {-0.6f, -0.125f, 0.59f, 0.05f, 0.5f, RGBSpectrum(0.9f, 0.7f, 0.9f)},
        {0.0f, -0.125f, 0.59f, 0.2f, 0.5f, RGBSpectrum(0.4f, 0.7f, 0.9f)},  {0.6f, -0.125f, 0.59f, 0.5f, 0.5f, RGBSpectrum(0.4f, 0.9f, 0.9f)},
        {-0.6f, 0.5f, 0.09f, 0.05f, 0.1f, RGBSpectrum(0.9f, 0.7f, 0.9f)},   {0.0f, 0.5f, 0.09f, 0.2f, 0.1f, RGBSpectrum(0.4f, 0.9f, 0.9f)},
        {0.6f, 0.5f, 0.09f, 0.5f, 0.1f, RGBSpectrum(0.4f, 0.7f, 0.9f)},
    };
    for (const Sp &p : sp) // scene.rs:219-308
        s.world.objects.push(Sphere(Vec3(p.x, p.y, -2.5f), 0.25f, std::make_shared<Microfacet>(RGBSpectrum(p.f0, p.f0, p.f0), p.rough, p.metal, p.att)));
    push_ceiling_light(s.world);
    return s;
}

// scene.rs:356-401
Scene Scene::blue_noise_test(float aspect) {
    Scene s;
    FastMesh board(make_chess_board(Pt3(-1, 0, -1), Pt3(1, 0, -1), Pt3(1, 0, -3), Pt3(-1, 0, -3), 24));
    board.displacement(Vec3(0.0f, 0.0f, 2.0f));
    board.scale(5.0f);
    board.rotate(10.0f, 30.0f, 5.1f);
    board.displacement(Vec3(0.0f, -0.5f, -7.0f));
    s.cam = cornell_camera(aspect);
    s.world.objects.push(std::move(board));
    s.world.lights.push(PointLight(Pt3(0.0f, 2.0f, -5.0f), RGBSpectrum(0.9f, 0.64f, 0.48f) * 10.8f));
    return s;
}

// scene.rs:403-552
Scene Scene::light_test(SamplerKind, float aspect) {
    Scene s;
    const float ceiling = -0.5f, ground = -0.57f, brightness = 8.0f;
    auto fl = make_square(Pt3(-1.0f, ground, -1.0f), Pt3(1.0f, ground, -1.0f), Pt3(1.0f, ground, -3.0f), Pt3(-1.0f, ground, -3.0f), RGBSpectrum(1.0f, 1.0f, 1.0f));
    MaterialRef dummy = std::make_shared<Dielectric>(0.0f);
    NaiveMesh triangle(std::vector<Triangle>{Triangle(Pt3(-0.15f, ceiling, -2.15f), Pt3(0.15f, ceiling, -2.15f), Pt3(0.0f, ceiling, -1.85f), dummy)});
    const float interval = 0.6f;
    auto sq = make_square(Pt3(-0.1f + interval, ceiling, -1.9f), Pt3(0.1f + interval, ceiling, -1.9f), Pt3(0.1f + interval, ceiling, -2.1f),
                          Pt3(-0.1f + interval, ceiling, -2.1f), BLACK);
    NaiveMesh square(std::vector<Triangle>{sq.first, sq.second});
    auto P = [&](float x, float z) { return Pt3(x - interval, ceiling, z); };
    NaiveMesh hexagon(std::vector<Triangle>{
        Triangle(P(-0.1f, -2.17f), P(0.1f, -2.17f), P(0.0f, -2.0f), dummy), Triangle(P(0.1f, -2.17f), P(0.2f, -2.0f), P(0.0f, -2.0f), dummy),
        Triangle(P(-0.2f, -2.0f), P(-0.1f, -2.17f), P(0.0f, -2.0f), dummy), Triangle(P(0.1f, -1.83f), P(-0.1f, -1.83f), P(0.0f, -2.0f), dummy),
        Triangle(P(0.2f, -2.0f), P(0.1f, -1.83f), P(0.0f, -2.0f), dummy),   Triangle(P(-0.1f, -1.83f), P(-0.2f, -2.0f), P(0.0f, -2.0f), dummy)});
    s.cam = Camera(Pt3(0.0f, 1.5f, 3.0f), Pt3(0.0f, 0.0f, -1.0f), Vec3(0.0f, 1.0f, 0.0f), 90.0f, aspect);
    s.world.objects.push(fl.first);
    s.world.objects.push(fl.second);
    RGBSpectrum sp = RGBSpectrum(0.9f, 0.64f, 0.48f) * brightness;
    s.world.lights.push(PolygonLight(std::move(triangle), sp));
    s.world.lights.push(PolygonLight(std::move(square), sp));
    s.world.lights.push(PolygonLight(std::move(hexagon), sp));
    return s;
}

// scene.rs:554-699
Scene Scene::mis_test(SamplerKind k, float aspect) {
    Scene s;
    const float ceiling = 2.0f, ground = -1.0f, width = 2.0f, front = -1.0f, back = -4.0f;
    Pt3 v3(width, ground, back), v4(-width, ground, back);
    auto a = make_square(Pt3(-width, ground, front), Pt3(width, ground, front), v3, v4, RGBSpectrum(0.6f, 0.6f, 0.6f));
    Pt3 v7(width, ceiling, back), v8(-width, ceiling, back);
    auto b = make_square(v4, v3, v7, v8, RGBSpectrum(0.6f, 0.6f, 0.6f));
    const float bw = 1.2f;
    auto bar = [&](float y0, float z0, float y1, float z1, float rough, float metal) {
        return make_mf_square(Pt3(-bw, ground + y0, z0), Pt3(bw, ground + y0, z0), Pt3(bw, ground + y1, z1), Pt3(-bw, ground + y1, z1),
                              Microfacet(RGBSpectrum(0.9f, 0.9f, 0.9f), rough, metal, RGBSpectrum(0.3f, 0.3f, 0.3f)));
    };
    auto b1 = bar(0.07f, -1.6f, 0.16f, -2.1f, 0.3f, 0.9f);
    auto b2 = bar(0.16f, -2.1f - 0.25f, 0.295f, -2.75f, 0.25f, 0.9f);
    auto b3 = bar(0.3f, -2.75f - 0.25f, 0.5f, -3.35f, 0.2f, 0.1f);
    auto b4 = bar(0.5f, -3.45f - 0.2f, 0.775f, -3.9f, 0.1f, 0.9f);
    s.cam = Camera(Pt3(0.0f, 1.0f, 3.0f), Pt3(0.0f, 0.0f, -1.0f), Vec3(0.0f, 1.0f, 0.0f), 90.0f, aspect);
    for (auto *p : {&a, &b, &b1, &b2, &b3, &b4}) {
        s.world.objects.push(p->first);
        s.world.objects.push(p->second);
    }
    const float interval = 0.7f, st = interval / 2.0f;
    s.world.lights.push(DiskLight(Pt3(st - interval * 2.0f, 0.5f, -3.5f), 0.02f, RGBSpectrum(0.9f, 0.4f, 0.6f) * 150.0f, k));
    s.world.lights.push(DiskLight(Pt3(st - interval, 0.5f, -3.5f), 0.05f, RGBSpectrum(0.9f, 0.7f, 0.3f) * 60.0f, k));
    s.world.lights.push(DiskLight(Pt3(st, 0.5f, -3.5f), 0.1f, RGBSpectrum(0.6f, 0.9f, 0.4f) * 40.0f, k));
    s.world.lights.push(DiskLight(Pt3(st + interval, 0.5f, -3.5f), 0.2f, RGBSpectrum(0.4f, 0.6f, 0.9f) * 20.0f, k));
    return s;
}

// ---- benchmark scenes ------------------------------------------------------------------------------------------
namespace {
FastMesh placed_utah(const std::string &assets, MaterialRef m) { // scene.rs:110-118
    FastMesh utah = load_obj_file_ref(assets + "/utah.obj", std::move(m));
    utah.scale(0.1f);
    utah.displacement(Vec3(0.1f, -0.507f, -1.9f));
    return utah;
}
FastMesh placed_pyramid(const std::string &assets, MaterialRef m) { // scene.rs:65-80
    FastMesh p = load_obj_file_ref(assets + "/pyramid.obj", std::move(m));
    p.scale(8.0f);
    p.rotate(0.0f, -15.0f, 0.0f);
    p.displacement(Vec3(0.55f, 0.1f, -2.6f));
    return p;
}
// scene.rs:87-100 placed `.miku2.obj`, which is not in the repository; input/miku.obj stands in with the same transform
FastMesh placed_miku_right(const std::string &assets, MaterialRef m) {
    FastMesh k = load_obj_file_ref(assets + "/miku.obj", std::move(m));
    k.transform(0.008f, Vec3(0.6f, -1.0f, -1.3f), -90.0f, 0.0f, -35.0f);
    return k;
}
// scene.rs:102-109 placed `.miku3.obj` (absent); input/miku.obj stands in, at miku.obj's own scale
FastMesh placed_miku_left(const std::string &assets, MaterialRef m) {
    FastMesh k = load_obj_file_ref(assets + "/miku.obj", std::move(m));
    k.transform(0.008f, Vec3(-0.5f, -1.0f, -1.5f), -90.0f, 0.0f, 35.0f);
    return k;
}
} // namespace

// The scene of output/cornell.jpg: box, ceiling polygon light, metal sphere, metal cylinder under a diffuse teapot,
// mirror pyramid, a glass figure on the left and a metal figure on the right (scene.rs:65-118,173-218 comments).
Scene Scene::bench_cornell(SamplerKind, float aspect, const std::string &assets) {
    Scene s;
    s.cam = cornell_camera(aspect);
    s.world.objects.push(placed_utah(assets, std::make_shared<Diffuse>(RGBSpectrum(1.0f, 1.0f, 1.0f))));
    s.world.objects.push(placed_miku_right(assets, std::make_shared<Metal>(RGBSpectrum(1.0f, 0.7f, 0.9f))));
    s.world.objects.push(placed_miku_left(assets, std::make_shared<Dielectric>(1.8f)));
    s.world.objects.push(placed_pyramid(assets, std::make_shared<Metal>(RGBSpectrum(1.0f, 1.0f, 1.0f))));
    push_cornell_walls(s.world);
    s.world.objects.push(Cylinder(0.1f, -1.95f, -1.0f, -0.5f, 0.4f, std::make_shared<Metal>(RGBSpectrum(0.9f, 0.7f, 0.4f))));
    s.world.objects.push(Sphere(Vec3(-0.34f, 0.38f, -2.0f), 0.25f, std::make_shared<Metal>(RGBSpectrum(0.4f, 0.7f, 0.9f))));
    push_ceiling_light(s.world);
    return s;
}

// The scene family of output/final.jpg: Whitted shader, disk light of scene.rs:327-332, cube.obj and pyramid.obj
// with Metal / Dielectric / Diffuse materials.  cube.obj has only axis-aligned faces; it is rendered intact
// (brute-force mesh semantics), where the reference's BVH would drop faces whose leaf box is flat (mesh.rs:117).
Scene Scene::bench_whitted(SamplerKind k, float aspect, const std::string &assets) {
    Scene s;
    s.cam = cornell_camera(aspect);
    FastMesh cube = load_obj_file_ref(assets + "/cube.obj", std::make_shared<Metal>(RGBSpectrum(0.9f, 0.7f, 0.4f)));
    cube.scale(0.1f);
    cube.rotate(0.0f, 25.0f, 0.0f);
    cube.displacement(Vec3(-0.75f, -1.0f, -2.3f));
    s.world.objects.push(std::move(cube));
    s.world.objects.push(placed_pyramid(assets, std::make_shared<Dielectric>(1.8f)));
    push_cornell_walls(s.world);
    s.world.objects.push(Cylinder(0.35f, -1.85f, -1.0f, -0.6f, 0.3f, std::make_shared<Diffuse>(RGBSpectrum(0.8f, 0.8f, 0.8f))));
    s.world.objects.push(Sphere(Vec3(-0.34f, 0.38f, -2.0f), 0.25f, std::make_shared<Metal>(RGBSpectrum(0.4f, 0.7f, 0.9f))));
    s.world.objects.push(Sphere(Vec3(0.35f, -0.35f, -1.85f), 0.25f, std::make_shared<Dielectric>(1.5f)));
    s.world.lights.push(DiskLight(Pt3(0.0f, 1.0f, -2.0f), 0.4f, RGBSpectrum(0.9f, 0.64f, 0.48f) * 6.0f, k));
    return s;
}

// The scene of output/sampling/cornell/{1light,2light}/*_16spp.jpg (and of output/path1.jpg, sampling/mis/{brdf,mis}_*spp.jpg):
// the box with the chess board of scene.rs:136-142 as its back wall, the objects of scene.rs:173-218 (comments) in their
// Whitted-compatible materials, and the disk lights of scene.rs:327-344: one warm light in the middle of the ceiling,
// or the magenta and the cyan one.  The two figure meshes of those renders are not in the repository and are left out.
Scene Scene::cornell_sampling(SamplerKind k, float aspect, const std::string &assets, int n_lights) {
    Scene s;
    s.cam = cornell_camera(aspect);
    s.world.objects.push(placed_utah(assets, std::make_shared<Diffuse>(RGBSpectrum(1.0f, 1.0f, 1.0f))));
    s.world.objects.push(placed_pyramid(assets, std::make_shared<Metal>(RGBSpectrum(1.0f, 1.0f, 1.0f))));
    Pt3 v1(-1, -1, -1), v2(1, -1, -1), v3(1, -1, -3), v4(-1, -1, -3);
    Pt3 v5(-1, 1, -1), v6(1, 1, -1), v7(1, 1, -3), v8(-1, 1, -3);
    auto floor = make_square(v1, v2, v3, v4, RGBSpectrum(0.9f, 0.9f, 0.9f));
    auto ceiling = make_square(v5, v8, v7, v6, RGBSpectrum(0.9f, 0.9f, 0.9f));
    s.world.objects.push(floor.first); s.world.objects.push(floor.second);
    s.world.objects.push(ceiling.first); s.world.objects.push(ceiling.second);
    FastMesh board(make_chess_board(v4, v3, v7, v8, 12)); // scene.rs:136-142
    board.rotate(0.01f, 0.0f, 0.0f);
    s.world.objects.push(std::move(board));
    auto left = make_square(v1, v4, v8, v5, RGBSpectrum(0.7f, 0.0f, 0.0f));
    auto right = make_square(v3, v2, v6, v7, RGBSpectrum(0.0f, 0.4f, 0.0f));
    s.world.objects.push(left.first); s.world.objects.push(left.second);
    s.world.objects.push(right.first); s.world.objects.push(right.second);
    s.world.objects.push(Cylinder(0.1f, -1.95f, -1.0f, -0.5f, 0.4f, std::make_shared<Metal>(RGBSpectrum(0.9f, 0.7f, 0.4f))));
    s.world.objects.push(Sphere(Vec3(-0.34f, 0.38f, -2.0f), 0.25f, std::make_shared<Metal>(RGBSpectrum(0.4f, 0.7f, 0.9f))));
    if (n_lights == 1) {
        s.world.lights.push(DiskLight(Pt3(0.0f, 1.0f, -2.0f), 0.4f, RGBSpectrum(0.9f, 0.64f, 0.48f) * 6.0f, k));  // scene.rs:327-332
    } else {
        s.world.lights.push(DiskLight(Pt3(-0.45f, 1.0f, -2.25f), 0.4f, RGBSpectrum(0.9f, 0.0f, 0.48f) * 6.0f, k)); // scene.rs:333-338
        s.world.lights.push(DiskLight(Pt3(0.45f, 1.0f, -1.75f), 0.4f, RGBSpectrum(0.0f, 0.63f, 0.48f) * 6.0f, k)); // scene.rs:339-344
    }
    return s;
}

// Box + ceiling light + Utah teapot with the Microfacet of scene.rs:71-76.
Scene Scene::bench_utah(SamplerKind, float aspect, const std::string &assets) {
    Scene s;
    s.cam = cornell_camera(aspect);
    s.world.objects.push(placed_utah(assets, std::make_shared<Microfacet>(mf_pyramid())));
    push_cornell_walls(s.world);
    push_ceiling_light(s.world);
    return s;
}

// Box + ceiling light + miku.obj with the Microfacet of scene.rs:93-98.
Scene Scene::bench_miku(SamplerKind, float aspect, const std::string &assets) {
    Scene s;
    s.cam = cornell_camera(aspect);
    s.world.objects.push(placed_miku_right(assets, std::make_shared<Microfacet>(mf_miku())));
    push_cornell_walls(s.world);
    push_ceiling_light(s.world);
    return s;
}

// The composite of output/final3.jpg with the materials of scene.rs:71-76,93-98,104,113-115,198-203,212-217.
Scene Scene::bench_composite(SamplerKind, float aspect, const std::string &assets) {
    Scene s;
    s.cam = cornell_camera(aspect);
    s.world.objects.push(placed_utah(assets, std::make_shared<Diffuse>(RGBSpectrum(1.0f, 1.0f, 1.0f))));
    s.world.objects.push(placed_miku_right(assets, std::make_shared<Microfacet>(mf_miku())));
    s.world.objects.push(placed_miku_left(assets, std::make_shared<Dielectric>(1.8f)));
    s.world.objects.push(placed_pyramid(assets, std::make_shared<Microfacet>(mf_pyramid())));
    push_cornell_walls(s.world);
    s.world.objects.push(Cylinder(0.1f, -1.95f, -1.0f, -0.5f, 0.4f, std::make_shared<Microfacet>(mf_cylinder())));
    s.world.objects.push(Sphere(Vec3(-0.34f, 0.38f, -2.0f), 0.25f, std::make_shared<Microfacet>(mf_sphere())));
    push_ceiling_light(s.world);
    return s;
}

} // namespace rt

// ==================================================================================================== C entry points
struct rth_scene {
    std::unique_ptr<rt::FlatScene> flat;
    rt_camera cam;
};

static thread_local std::string g_host_err;

extern "C" {

const char *rth_last_error(void) { return g_host_err.c_str(); }

int32_t rth_scene_create(const char *name, float aspect, int32_t light_sampler, const char *asset_dir, rth_scene **out) {
    if (!name || !out) { g_host_err = "null argument"; return RT_ERR_INVALID; }
    try {
        rt::SamplerKind k = rt::sampler_kind_from_int(light_sampler);
        std::string assets = asset_dir ? asset_dir : "assets";
        std::string n = name;
        rt::Scene s;
        if (n == "head") s = rt::Scene::cornell_box(k, aspect);
        else if (n == "blue_noise_test") s = rt::Scene::blue_noise_test(aspect);
        else if (n == "light_test") s = rt::Scene::light_test(k, aspect);
        else if (n == "mis_test") s = rt::Scene::mis_test(k, aspect);
        else if (n == "cornell") s = rt::Scene::bench_cornell(k, aspect, assets);
        else if (n == "whitted") s = rt::Scene::bench_whitted(k, aspect, assets);
        else if (n == "cornell_1light") s = rt::Scene::cornell_sampling(k, aspect, assets, 1);
        else if (n == "cornell_2light") s = rt::Scene::cornell_sampling(k, aspect, assets, 2);
        else if (n == "utah") s = rt::Scene::bench_utah(k, aspect, assets);
        else if (n == "miku") s = rt::Scene::bench_miku(k, aspect, assets);
        else if (n == "composite") s = rt::Scene::bench_composite(k, aspect, assets);
        else { g_host_err = "unknown scene: " + n; return RT_ERR_INVALID; }
        rth_scene *h = new rth_scene();
        h->flat = s.flatten();
        h->cam = s.cam.describe();
        *out = h;
        return RT_OK;
    } catch (const std::exception &e) {
        g_host_err = e.what();
        return RT_ERR_INVALID;
    }
}

void rth_scene_destroy(rth_scene *s) { delete s; }
const rt_scene_desc *rth_scene_desc(const rth_scene *s) { return &s->flat->desc; }
const rt_camera *rth_scene_camera(const rth_scene *s) { return &s->cam; }

void rth_camera_new(const float *f, const float *a, const float *u, float vfov, float aspect, rt_camera *out) {
    *out = rt::Camera(rt::Pt3(f[0], f[1], f[2]), rt::Pt3(a[0], a[1], a[2]), rt::Vec3(u[0], u[1], u[2]), vfov, aspect).describe();
}

void rth_transform_points(float *pts, int32_t n, float scale, const float *disp, float x, float y, float z) {
    rt::Quat q = rt::quat_from_euler_deg(x, y, z);
    for (int i = 0; i < n; i++) {
        rt::Vec3 p = rt::transform_point(q, scale, rt::Vec3(disp[0], disp[1], disp[2]), rt::Vec3(pts[3 * i], pts[3 * i + 1], pts[3 * i + 2]));
        pts[3 * i] = p.x; pts[3 * i + 1] = p.y; pts[3 * i + 2] = p.z;
    }
}

int32_t rth_load_obj(const char *path, float **tri_v) {
    try {
        std::vector<float> tri = rt::parse_obj(path);
        float *p = (float *)std::malloc(tri.size() * sizeof(float) + 4);
        if (!p) { g_host_err = "out of memory"; return -RT_ERR_NOMEM; }
        std::memcpy(p, tri.data(), tri.size() * sizeof(float));
        *tri_v = p;
        return (int32_t)(tri.size() / 9);
    } catch (const std::exception &e) {
        g_host_err = e.what();
        return -RT_ERR_INVALID;
    }
}
void rth_free(void *p) { std::free(p); }

int32_t rth_write_ppm(const char *path, int32_t w, int32_t h, const int32_t *rgb) {
    FILE *f = std::fopen(path, "w");
    if (!f) { g_host_err = std::string("cannot create ") + path; return RT_ERR_INVALID; }
    std::fprintf(f, "P3\n%d %d\n255\n", w, h); // main.rs:27
    for (int64_t i = 0; i < (int64_t)w * h; i++) std::fprintf(f, "%d %d %d\n", rgb[3 * i], rgb[3 * i + 1], rgb[3 * i + 2]); // main.rs:51
    std::fclose(f);
    return RT_OK;
}

namespace {
// CRC-32 of PNG chunks (polynomial 0xEDB88320), table built on first use
uint32_t png_crc(uint32_t crc, const uint8_t *p, size_t n) {
    static uint32_t table[256];
    static bool ready = false;
    if (!ready) {
        for (uint32_t i = 0; i < 256; i++) {
            uint32_t c = i;
            for (int k = 0; k < 8; k++) c = (c & 1u) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
            table[i] = c;
        }
        ready = true;
    }
    crc = ~crc;
    for (size_t i = 0; i < n; i++) crc = table[(crc ^ p[i]) & 0xFFu] ^ (crc >> 8);
    return ~crc;
}
void put_be32(std::vector<uint8_t> &v, uint32_t x) { v.push_back(x >> 24); v.push_back(x >> 16); v.push_back(x >> 8); v.push_back(x); }
bool png_chunk(FILE *f, const char type[4], const std::vector<uint8_t> &data) {
    std::vector<uint8_t> head;
    put_be32(head, (uint32_t)data.size());
    head.insert(head.end(), type, type + 4);
    uint32_t crc = png_crc(0u, head.data() + 4, 4);
    crc = png_crc(crc, data.data(), data.size());
    std::vector<uint8_t> tail;
    put_be32(tail, crc);
    return std::fwrite(head.data(), 1, 8, f) == 8 && (data.empty() || std::fwrite(data.data(), 1, data.size(), f) == data.size()) &&
           std::fwrite(tail.data(), 1, 4, f) == 4;
}
} // namespace

// resize.py:8-11: pixel (i, j) of the enlarged image is pixel (i / rate, j / rate) of the source
int32_t rth_upscale_nearest(int32_t w, int32_t h, int32_t rate, const int32_t *rgb, int32_t *out) {
    if (!rgb || !out || w <= 0 || h <= 0 || rate <= 0 || (int64_t)w * rate > 65535 || (int64_t)h * rate > 65535) { g_host_err = "rth_upscale_nearest: bad argument"; return RT_ERR_INVALID; }
    const int64_t W = (int64_t)w * rate;
    for (int64_t y = 0; y < (int64_t)h * rate; y++) {
        const int32_t *src = rgb + (y / rate) * w * 3;
        int32_t *dst = out + y * W * 3;
        for (int64_t x = 0; x < W; x++) {
            const int32_t *s = src + (x / rate) * 3;
            dst[3 * x] = s[0]; dst[3 * x + 1] = s[1]; dst[3 * x + 2] = s[2];
        }
    }
    return RT_OK;
}

int32_t rth_write_png(const char *path, int32_t w, int32_t h, const int32_t *rgb) {
    if (!path || !rgb || w <= 0 || h <= 0) { g_host_err = "rth_write_png: bad argument"; return RT_ERR_INVALID; }
    FILE *f = std::fopen(path, "wb");
    if (!f) { g_host_err = std::string("cannot create ") + path; return RT_ERR_INVALID; }
    // rows of (filter byte 0, RGB bytes); values above 255 (the reference's cast does not clamp, main.rs:48-50) saturate
    const size_t row = 1 + (size_t)w * 3;
    std::vector<uint8_t> raw(row * (size_t)h);
    for (int32_t y = 0; y < h; y++) {
        uint8_t *r = raw.data() + row * y;
        r[0] = 0;
        for (int64_t i = 0; i < (int64_t)w * 3; i++) {
            const int32_t v = rgb[(int64_t)y * w * 3 + i];
            r[1 + i] = (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v);
        }
    }
    // zlib stream of stored blocks (deflate type 0, at most 65535 bytes each) with its Adler-32
    std::vector<uint8_t> z;
    z.reserve(raw.size() + raw.size() / 65535 * 5 + 16);
    z.push_back(0x78); z.push_back(0x01);
    uint32_t a = 1, b = 0;
    for (size_t pos = 0; pos < raw.size() || pos == 0;) {
        const size_t n = std::min<size_t>(65535, raw.size() - pos);
        z.push_back(pos + n >= raw.size() ? 1 : 0);
        z.push_back(n & 0xFF); z.push_back(n >> 8); z.push_back(~n & 0xFF); z.push_back((~n >> 8) & 0xFF);
        z.insert(z.end(), raw.begin() + pos, raw.begin() + pos + n);
        for (size_t i = pos; i < pos + n; i++) { a = (a + raw[i]) % 65521u; b = (b + a) % 65521u; }
        pos += n;
        if (n == 0) break;
    }
    put_be32(z, (b << 16) | a);
    std::vector<uint8_t> ihdr;
    put_be32(ihdr, (uint32_t)w); put_be32(ihdr, (uint32_t)h);
    ihdr.push_back(8); ihdr.push_back(2); ihdr.push_back(0); ihdr.push_back(0); ihdr.push_back(0); // 8 bits, RGB, no interlace
    static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', '\r', '\n', 0x1A, '\n'};
    bool ok = std::fwrite(sig, 1, 8, f) == 8 && png_chunk(f, "IHDR", ihdr) && png_chunk(f, "IDAT", z) && png_chunk(f, "IEND", {});
    ok = (std::fclose(f) == 0) && ok;
    if (!ok) { g_host_err = std::string("cannot write ") + path; return RT_ERR_INVALID; }
    return RT_OK;
}

namespace {
struct AccumHeader {
    char magic[8];
    int32_t width, height, spp, spp_done;
    uint32_t seed, tag; // tag: what else the sums depend on (scene, shader, samplers ...), 0 = not recorded
};
static_assert(sizeof(AccumHeader) == 32, "accumulation file header is 32 bytes");
} // namespace

uint32_t rth_render_tag(const char *description) { // FNV-1a; never 0
    uint32_t hsh = 2166136261u;
    for (const char *c = description ? description : ""; *c; c++) hsh = (hsh ^ (uint32_t)(unsigned char)*c) * 16777619u;
    return hsh ? hsh : 1u;
}

int32_t rth_accum_save(const char *path, int32_t w, int32_t h, int32_t spp, int32_t spp_done, uint32_t seed, const float *accum) {
    return rth_accum_save_tagged(path, w, h, spp, spp_done, seed, 0u, accum);
}

int32_t rth_accum_load(const char *path, int32_t w, int32_t h, int32_t spp, uint32_t seed, int32_t *spp_done, float *accum) {
    return rth_accum_load_tagged(path, w, h, spp, seed, 0u, spp_done, accum);
}

int32_t rth_accum_save_tagged(const char *path, int32_t w, int32_t h, int32_t spp, int32_t spp_done, uint32_t seed, uint32_t tag, const float *accum) {
    if (!path || !accum || w <= 0 || h <= 0 || spp <= 0 || spp_done < 0 || spp_done > spp) { g_host_err = "rth_accum_save: bad argument"; return RT_ERR_INVALID; }
    // written under a temporary name and renamed: a run killed while saving keeps the previous state
    const std::string tmp = std::string(path) + ".tmp";
    FILE *f = std::fopen(tmp.c_str(), "wb");
    if (!f) { g_host_err = std::string("cannot create ") + tmp; return RT_ERR_INVALID; }
    AccumHeader hd{{'R', 'T', 'A', 'C', 'C', 'U', 'M', '1'}, w, h, spp, spp_done, seed, tag};
    const size_t n = (size_t)w * (size_t)h * 3;
    bool ok = std::fwrite(&hd, sizeof(hd), 1, f) == 1 && std::fwrite(accum, sizeof(float), n, f) == n;
    ok = (std::fclose(f) == 0) && ok;
    if (!ok || std::rename(tmp.c_str(), path) != 0) { g_host_err = std::string("cannot write ") + path; std::remove(tmp.c_str()); return RT_ERR_INVALID; }
    return RT_OK;
}

int32_t rth_accum_load_tagged(const char *path, int32_t w, int32_t h, int32_t spp, uint32_t seed, uint32_t tag, int32_t *spp_done, float *accum) {
    if (!path || !accum || !spp_done) { g_host_err = "rth_accum_load: bad argument"; return RT_ERR_INVALID; }
    FILE *f = std::fopen(path, "rb");
    if (!f) { g_host_err = std::string("cannot open ") + path; return RT_ERR_INVALID; }
    AccumHeader hd;
    const size_t n = (size_t)w * (size_t)h * 3;
    int32_t rc = RT_OK;
    if (std::fread(&hd, sizeof(hd), 1, f) != 1 || std::memcmp(hd.magic, "RTACCUM1", 8) != 0) { g_host_err = std::string(path) + ": not an accumulation file"; rc = RT_ERR_INVALID; }
    else if (hd.width != w || hd.height != h || hd.spp != spp || hd.seed != seed) { g_host_err = std::string(path) + ": saved for another frame size, sample count or seed"; rc = RT_ERR_INVALID; }
    else if (hd.tag != tag) { g_host_err = std::string(path) + ": saved by another render (scene, shader, samplers, light samples or depth differ)"; rc = RT_ERR_INVALID; }
    else if (hd.spp_done < 0 || hd.spp_done > hd.spp) { g_host_err = std::string(path) + ": bad sample count"; rc = RT_ERR_INVALID; }
    else if (std::fread(accum, sizeof(float), n, f) != n) { g_host_err = std::string(path) + ": truncated"; rc = RT_ERR_INVALID; }
    else *spp_done = hd.spp_done;
    std::fclose(f);
    return rc;
}

} // extern "C"
