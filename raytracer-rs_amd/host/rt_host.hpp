// rt_host.hpp — host-side mirror of the reference's scene API.
//
// Same concepts and names as raytracer-rs (Scene, World, HitableList, Triangle, Sphere, Cylinder, Mesh<Accel>,
// Material, Light, Camera, SamplerKind, load_obj_file), but the objects only DESCRIBE the scene: intersection,
// shading and sampling run on the GPU.  Scene::flatten() turns a scene into the rt_scene_desc of rt_api.h
// (SoA class arrays in world order), which rt_scene_create uploads.
#pragma once

#include "../../include/rt_api.h"

#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

namespace rt {

// src/lib.rs:12-16
struct Vec3 {
    float x = 0, y = 0, z = 0;
    Vec3() = default;
    Vec3(float x_, float y_, float z_) : x(x_), y(y_), z(z_) {}
};
using Pt3 = Vec3;
using RGBSpectrum = Vec3;

inline Vec3 operator+(Vec3 a, Vec3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline Vec3 operator-(Vec3 a, Vec3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline Vec3 operator*(Vec3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
inline Vec3 operator*(float s, Vec3 a) { return {s * a.x, s * a.y, s * a.z}; }
inline Vec3 operator/(Vec3 a, float s) { return {a.x / s, a.y / s, a.z / s}; }
float dot(Vec3 a, Vec3 b);
Vec3 cross(Vec3 a, Vec3 b);
Vec3 normalize(Vec3 a);

// src/light.rs:14-15
const RGBSpectrum BLACK{0.0f, 0.0f, 0.0f};
const RGBSpectrum WHITE{1.0f, 1.0f, 1.0f};

// src/sampler.rs:5-22
enum class SamplerKind { WhiteNoise = RT_SAMPLER_WHITE, Uniform = RT_SAMPLER_UNIFORM, Jittered = RT_SAMPLER_JITTERED, BlueNoise = RT_SAMPLER_BLUE };
SamplerKind sampler_kind_from_int(int i); // throws std::invalid_argument("unexpected integer")

// ---- materials (src/material.rs:31-34,117-120,182-185,305-311) ------------------------------------------------
struct Material {
    virtual ~Material() = default;
    virtual rt_material describe() const = 0;
};
struct Diffuse : Material {
    RGBSpectrum albedo;
    explicit Diffuse(RGBSpectrum a) : albedo(a) {}
    rt_material describe() const override;
};
struct Metal : Material {
    RGBSpectrum albedo;
    explicit Metal(RGBSpectrum a) : albedo(a) {}
    rt_material describe() const override;
};
struct Dielectric : Material {
    float ref_idx;
    explicit Dielectric(float r) : ref_idx(r) {}
    rt_material describe() const override;
};
struct Microfacet : Material {
    RGBSpectrum f0;
    float roughness, metallic;
    RGBSpectrum attenuation;
    Microfacet(RGBSpectrum f0_, float roughness_, float metallic_, RGBSpectrum att) : f0(f0_), roughness(roughness_), metallic(metallic_), attenuation(att) {}
    rt_material describe() const override;
};
using MaterialRef = std::shared_ptr<const Material>; // Rc<dyn Material>

// ---- hitables (src/hitable.rs:21-27, src/geometry.rs:9-12,76-80,119-126) ---------------------------------------
struct Flattener;
struct Hitable {
    virtual ~Hitable() = default;
    virtual void flatten(Flattener &f) const = 0;
};
struct Triangle : Hitable {
    Pt3 vertex[3];
    MaterialRef mat;
    Triangle(Pt3 a, Pt3 b, Pt3 c, MaterialRef m) : vertex{a, b, c}, mat(std::move(m)) {}
    Vec3 normal() const; // geometry.rs:16-20
    float area() const;  // geometry.rs:23-26
    void flatten(Flattener &f) const override;
};
struct Sphere : Hitable {
    Vec3 center;
    float radius;
    MaterialRef mat;
    Sphere(Vec3 c, float r, MaterialRef m) : center(c), radius(r), mat(std::move(m)) {}
    void flatten(Flattener &f) const override;
};
struct Cylinder : Hitable {
    float center_x, center_z, y_min, y_max, radius;
    MaterialRef mat;
    Cylinder(float cx, float cz, float ymin, float ymax, float r, MaterialRef m) : center_x(cx), center_z(cz), y_min(ymin), y_max(ymax), radius(r), mat(std::move(m)) {}
    void flatten(Flattener &f) const override;
};

// src/mesh.rs:17-58.  The acceleration structure is a tag: the device library builds its own BVH over all triangles.
struct Naive { static constexpr int kind = RT_ACCEL_NAIVE; };
struct BoundingBox { static constexpr int kind = RT_ACCEL_BOX; };
struct BVHTree { static constexpr int kind = RT_ACCEL_BVH; };

struct MeshBase : Hitable {
    std::vector<Triangle> face_list;
    void transform(float scale, Vec3 disp, float x, float y, float z); // mesh.rs:23-43
    void scale(float s) { transform(s, Vec3(0, 0, 0), 0, 0, 0); }       // mesh.rs:45-47
    void displacement(Vec3 d) { transform(1.0f, d, 0, 0, 0); }          // mesh.rs:50-52
    void rotate(float x, float y, float z) { transform(1.0f, Vec3(0, 0, 0), x, y, z); } // mesh.rs:55-57
    virtual int accel_kind() const = 0;
    void flatten(Flattener &f) const override;
};
template <class Accel> struct Mesh : MeshBase {
    Mesh() = default;
    explicit Mesh(std::vector<Triangle> faces) { face_list = std::move(faces); } // FromFaceList::from_face_list
    int accel_kind() const override { return Accel::kind; }
};
using NaiveMesh = Mesh<Naive>;      // mesh.rs:62
using FastMesh = Mesh<BVHTree>;     // mesh.rs:98
using BoxMesh = Mesh<BoundingBox>;  // mesh.rs:99

struct HitableList { // hitable.rs:25-27
    std::vector<std::shared_ptr<Hitable>> list;
    template <class T> void push(T obj) { list.push_back(std::make_shared<T>(std::move(obj))); }
};

// ---- lights (src/light.rs:54-57,106-112,242-247) ----------------------------------------------------------------
struct Light {
    virtual ~Light() = default;
    virtual void flatten(Flattener &f) const = 0;
};
struct PointLight : Light {
    Pt3 origin;
    RGBSpectrum spectrum;
    PointLight(Pt3 o, RGBSpectrum s) : origin(o), spectrum(s) {}
    void flatten(Flattener &f) const override;
};
struct DiskLight : Light {
    Pt3 origin;
    float radius;
    RGBSpectrum spectrum;
    SamplerKind sampler_kind;
    DiskLight(Pt3 o, float r, RGBSpectrum s, SamplerKind k) : origin(o), radius(r), spectrum(s), sampler_kind(k) {}
    void flatten(Flattener &f) const override;
};
struct PolygonLight : Light {
    NaiveMesh mesh;
    RGBSpectrum spectrum;
    PolygonLight(NaiveMesh m, RGBSpectrum s); // checks planarity like light.rs:252-269 (throws std::invalid_argument)
    void flatten(Flattener &f) const override;
};
struct LightList {
    std::vector<std::shared_ptr<Light>> list;
    template <class T> void push(T l) { list.push_back(std::make_shared<T>(std::move(l))); }
};

// src/shader.rs:16-19
struct World {
    HitableList objects;
    LightList lights;
};

// src/camera.rs:5-35
struct Camera {
    Pt3 origin, lower_left_corner;
    Vec3 horizontal, vertical;
    Camera() = default;
    Camera(Pt3 lookfrom, Pt3 lookat, Vec3 vup, float vfov, float aspect);
    rt_camera describe() const;
};

// Owns the arrays an rt_scene_desc points into.
struct FlatScene {
    std::vector<float> tri_v, sph, cyl, light_tri_v;
    std::vector<int32_t> tri_mat, sph_mat, cyl_mat;
    std::vector<rt_material> mats;
    std::vector<rt_light> lights;
    std::vector<rt_object> objects;
    rt_scene_desc desc{};
    void finish(); // points desc at the vectors
};

struct Flattener {
    FlatScene &out;
    std::vector<const Material *> seen;
    int material_index(const MaterialRef &m);
};

// src/scene.rs:14-17
struct Scene {
    Camera cam;
    World world;
    std::unique_ptr<FlatScene> flatten() const;

    // src/scene.rs:64,356,403,554.  `aspect` replaces the compile-time NX/NY of scene.rs:168.
    static Scene cornell_box(SamplerKind k, float aspect);
    static Scene blue_noise_test(float aspect);
    static Scene light_test(SamplerKind k, float aspect);
    static Scene mis_test(SamplerKind k, float aspect);
    // benchmark scenes assembled from the reference's walls, light, meshes and (commented) materials
    static Scene bench_cornell(SamplerKind k, float aspect, const std::string &assets);
    static Scene bench_whitted(SamplerKind k, float aspect, const std::string &assets);
    static Scene cornell_sampling(SamplerKind k, float aspect, const std::string &assets, int n_lights);
    static Scene bench_utah(SamplerKind k, float aspect, const std::string &assets);
    static Scene bench_miku(SamplerKind k, float aspect, const std::string &assets);
    static Scene bench_composite(SamplerKind k, float aspect, const std::string &assets);
};

// src/scene.rs:19-61
std::pair<Triangle, Triangle> make_square(Pt3 a, Pt3 b, Pt3 c, Pt3 d, RGBSpectrum albedo);
std::pair<Triangle, Triangle> make_mf_square(Pt3 a, Pt3 b, Pt3 c, Pt3 d, const Microfacet &m);
std::vector<Triangle> make_chess_board(Pt3 a, Pt3 b, Pt3 c, Pt3 d, int num);

// src/obj_loader.rs:16-47; throws std::runtime_error (I/O, parse, "Model should be triangulated first")
template <class M> FastMesh load_obj_file(const std::string &path, const M &mat);
FastMesh load_obj_file_ref(const std::string &path, MaterialRef mat);
template <class M> FastMesh load_obj_file(const std::string &path, const M &mat) { return load_obj_file_ref(path, std::make_shared<M>(mat)); }

} // namespace rt
