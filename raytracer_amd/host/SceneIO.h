// SceneIO.h — a scene file made of the reference's one-line text records
// (Primitives / Light / RTMaterial operator<< and >>, reference inc/RTObjects.h:39-57,267-296,648-660,
// inc/RTMaterial.h:40-57,250-280).  The reference leaves the file around those records to its outer
// application; the container below is this layer's own:
//
//   # comment
//   Camera cx cy cz dx dy dz
//   Material <RTMaterial record>          e.g.  Material Metal 0 0 0 0.8 0.6 0.2 0 0 0 0.3
//   Object <material index> <Primitives record>   e.g.  Object 0 Sphere 0 0 3 0 0.5
//   Light px py pz ax ay az dx dy dz sx sy sz pow       (the Light record itself)
//
// Materials are numbered in file order from 0.  Numbers are written with 9 significant digits, so a
// reload reproduces every float the GPU path sees; the records stay readable by the reference's scan().
#ifndef RTAMD_SCENEIO_H
#define RTAMD_SCENEIO_H

#include <istream>
#include <map>
#include <ostream>
#include <sstream>
#include <string>

#include "Camera.h"
#include "RayTracer.h"

namespace rtamd {

// New primitive of the class the type word names, filled by its scan(); nullptr for an unknown word
// (the counterpart of RTMaterialManager::deserializeMaterial for primitives).
inline Primitives *deserializePrimitive(std::istream &is, RTMaterial *material) {
    std::string type;
    is >> type;
    Primitives *p = nullptr;
    if (type == "Sphere") p = new SphereObject();
    else if (type == "Plane") p = new PlaneObject();
    else if (type == "Cube") p = new CubeObject();
    else if (type == "Polygon") p = new PolygonObject();
    else return nullptr;
    p->setMaterial(material);
    is >> *p;
    return p;
}

inline void saveScene(std::ostream &os, const SceneManager &scene, const Camera *camera = nullptr) {
    const auto old = os.precision(9);
    if (camera)
        os << "Camera " << camera->center().x() << ' ' << camera->center().y() << ' ' << camera->center().z() << ' '
           << camera->direction().x() << ' ' << camera->direction().y() << ' ' << camera->direction().z() << '\n';
    std::map<const RTMaterial *, int> index;
    for (const Primitives *p : scene.primitives()) {
        const RTMaterial *m = p->material();
        if (m && index.emplace(m, (int)index.size()).second) os << "Material " << *m << '\n';
    }
    for (const Primitives *p : scene.primitives())
        os << "Object " << (p->material() ? index[p->material()] : -1) << ' ' << *p << '\n';
    for (const Light *l : scene.lights()) os << *l << '\n';
    os.precision(old);
}

// Appends the file's objects and lights to `scene` (materials go to `materials`, the pose to
// `camera` when given).  Returns false and fills `error` on a malformed record; what was read
// before it stays in the scene.
inline bool loadScene(std::istream &is, SceneManager &scene, RTMaterialManager &materials, Camera *camera,
                      std::string *error = nullptr) {
    std::vector<RTMaterial *> mats;
    std::string line;
    int lineNo = 0;
    auto bad = [&](const std::string &what) {
        if (error) *error = "line " + std::to_string(lineNo) + ": " + what;
        return false;
    };
    while (std::getline(is, line)) {
        ++lineNo;
        std::istringstream ls(line);
        std::string word;
        if (!(ls >> word) || word[0] == '#') continue;
        if (word == "Camera") {
            double c[6];
            for (double &v : c) ls >> v;
            if (!ls) return bad("Camera needs six numbers");
            if (camera) {
                camera->setCenter(gm::IPoint3(c[0], c[1], c[2]));
                camera->setDirection(gm::IVec3f(c[3], c[4], c[5]));
            }
        } else if (word == "Material") {
            RTMaterial *m = materials.deserializeMaterial(ls);
            if (!m) return bad("unknown material type");
            if (!ls) return bad("short material record");
            mats.push_back(m);
        } else if (word == "Object") {
            int mi = -1;
            ls >> mi;
            if (!ls || mi < 0 || mi >= (int)mats.size()) return bad("material index out of range");
            Primitives *p = deserializePrimitive(ls, mats[(size_t)mi]);
            if (!p) return bad("unknown primitive type");
            if (!ls) {
                delete p;
                return bad("short primitive record");
            }
            scene.addObject(p);
        } else if (word == "Light") {
            Light *l = new Light();
            ls >> *l;
            if (!ls) {
                delete l;
                return bad("short light record");
            }
            scene.addLight(l);
        } else {
            return bad("unknown record '" + word + "'");
        }
    }
    return true;
}

} // namespace rtamd
#endif
