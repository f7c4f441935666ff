// Reader for the reference's ".ray" scene language (grammar and defaults follow
// src/parser/Parser.cpp:17-1172, tokens follow src/parser/Tokenizer.cpp:70-265
// and the keyword table src/parser/Token.cpp:116-184).
#include <cctype>
#include <cstdlib>
#include <fstream>
#include <list>
#include <map>
#include <sstream>
#include <stdexcept>

#include "scene.h"

namespace rta {
namespace {

enum Tok { T_EOF, T_IDENT, T_SCALAR, T_LPAREN, T_RPAREN, T_LBRACE, T_RBRACE, T_COMMA, T_EQUALS,
           T_SEMI, T_KEYWORD };

struct Token {
  Tok kind = T_EOF;
  std::string text;   // identifier / keyword spelling
  double value = 0.0;
  int line = 0;
};

const char* const kKeywords[] = {
    "ambient_light", "ambient", "aspectratio", "atten_angle", "bottom_radius", "box", "camera",
    "capped", "color", "colour", "cone", "constant_attenuation_coeff", "cylinder", "diffuse",
    "direction", "directional_light", "emissive", "faces", "false", "fall_rate", "fov",
    "gennormals", "height", "index", "linear_attenuation_coeff", "material", "materials", "map",
    "name", "normals", "point_light", "points", "polymesh", "position",
    "quadratic_attenuation_coeff", "quaternian", "reflective", "rotate", "SBT-raytracer", "scale",
    "shininess", "specular", "sphere", "spot_light", "square", "top_radius", "transform",
    "translate", "transmissive", "trimesh", "true", "updir", "viewdir"};

class Lexer {
 public:
  explicit Lexer(const std::string& s) : src_(s) {}

  Token next() {
    skipSpace();
    Token t;
    t.line = line_;
    if (pos_ >= src_.size()) return t;
    char c = src_[pos_];
    if (std::isalpha((unsigned char)c) || c == '_') {
      size_t b = pos_;
      while (pos_ < src_.size() &&
             (std::isalnum((unsigned char)src_[pos_]) || src_[pos_] == '_' || src_[pos_] == '-'))
        pos_++;
      t.text = src_.substr(b, pos_ - b);
      t.kind = T_IDENT;
      for (const char* k : kKeywords)
        if (t.text == k) { t.kind = T_KEYWORD; break; }
      if (t.text == "colour") t.text = "color";
      if (t.text == "polymesh") t.text = "trimesh";
      return t;
    }
    if (c == '"') {
      size_t b = ++pos_;
      while (pos_ < src_.size() && src_[pos_] != '"') {
        if (src_[pos_] == '\n') fail("Unterminated string constant");
        pos_++;
      }
      t.text = src_.substr(b, pos_ - b);
      pos_++;
      t.kind = T_IDENT;
      return t;
    }
    if (std::isdigit((unsigned char)c) || c == '-' || c == '.') {
      // Tokenizer.cpp:229-237: any run of [0-9.-e], converted with atof
      size_t b = pos_;
      while (pos_ < src_.size() && (std::isdigit((unsigned char)src_[pos_]) || src_[pos_] == '-' ||
                                    src_[pos_] == '.' || src_[pos_] == 'e'))
        pos_++;
      t.kind = T_SCALAR;
      t.value = std::atof(src_.substr(b, pos_ - b).c_str());
      return t;
    }
    pos_++;
    switch (c) {
      case '(': t.kind = T_LPAREN; break;
      case ')': t.kind = T_RPAREN; break;
      case '{': t.kind = T_LBRACE; break;
      case '}': t.kind = T_RBRACE; break;
      case ',': t.kind = T_COMMA; break;
      case '=': t.kind = T_EQUALS; break;
      case ';': t.kind = T_SEMI; break;
      default: fail(std::string("unexpected character: '") + c + "'");
    }
    return t;
  }

  [[noreturn]] void fail(const std::string& msg) const {
    std::ostringstream os;
    os << "Syntax error, line " << line_ << ": " << msg;
    throw std::runtime_error(os.str());
  }

 private:
  void skipSpace() {
    for (;;) {
      while (pos_ < src_.size() && std::isspace((unsigned char)src_[pos_])) {
        if (src_[pos_] == '\n') line_++;
        pos_++;
      }
      if (pos_ + 1 < src_.size() && src_[pos_] == '/' && src_[pos_ + 1] == '/') {
        while (pos_ < src_.size() && src_[pos_] != '\n') pos_++;
      } else if (pos_ + 1 < src_.size() && src_[pos_] == '/' && src_[pos_ + 1] == '*') {
        int start = line_;
        pos_ += 2;
        for (;;) {
          if (pos_ + 1 >= src_.size()) {
            std::ostringstream os;
            os << "Unterminated comment in line " << start;
            fail(os.str());
          }
          if (src_[pos_] == '*' && src_[pos_ + 1] == '/') { pos_ += 2; break; }
          if (src_[pos_] == '\n') line_++;
          pos_++;
        }
      } else {
        return;
      }
    }
  }
  const std::string& src_;
  size_t pos_ = 0;
  int line_ = 1;
};

class Parser {
 public:
  explicit Parser(const std::string& text) : lex_(text) { advance(); }

  std::unique_ptr<Scene> parseScene() {
    expectKeyword("SBT-raytracer");
    double version = scalar();
    if (version > 1.1) {
      std::ostringstream os;
      os << "SBT-raytracer version number " << version
         << " too high; only able to parse v1.1 and below.";
      throw std::runtime_error(os.str());
    }
    std::unique_ptr<Scene> scene(new Scene);
    Material mat;
    for (;;) {
      if (cur_.kind == T_EOF) return scene;
      if (cur_.kind == T_SEMI) { advance(); continue; }
      if (isGeometryStart()) {
        transformable(*scene, &scene->transformRoot, mat);
      } else if (isKeyword("point_light")) {
        pointLight(*scene);
      } else if (isKeyword("directional_light")) {
        directionalLight(*scene);
      } else if (isKeyword("spot_light")) {
        spotLight(*scene);
      } else if (isKeyword("ambient_light")) {
        advance();
        expect(T_LBRACE);
        if (!isKeyword("color")) lex_.fail("Expected color attribute");
        Vec3 c = vec3Expression();
        scene->ambientIntensity = add(scene->ambientIntensity, c);  // summed (Parser.cpp:719)
        expect(T_RBRACE);
      } else if (isKeyword("camera")) {
        camera(*scene);
      } else if (isKeyword("material")) {
        mat = materialExpression(mat);
      } else {
        lex_.fail("Expected: geometry, camera, or light information");
      }
    }
  }

 private:
  // ---- token helpers
  void advance() { cur_ = lex_.next(); }
  bool isKeyword(const char* k) const { return cur_.kind == T_KEYWORD && cur_.text == k; }
  void expect(Tok k) {
    if (cur_.kind != k) lex_.fail("unexpected token '" + cur_.text + "'");
    advance();
  }
  void expectKeyword(const char* k) {
    if (!isKeyword(k)) lex_.fail(std::string(k) + " expected");
    advance();
  }
  bool condRead(Tok k) {
    if (cur_.kind == k) { advance(); return true; }
    return false;
  }
  double scalar() {
    if (cur_.kind != T_SCALAR) lex_.fail("SCALAR expected");
    double v = cur_.value;
    advance();
    return v;
  }
  Vec3 vec3() {
    expect(T_LPAREN);
    double a = scalar(); expect(T_COMMA);
    double b = scalar(); expect(T_COMMA);
    double c = scalar(); expect(T_RPAREN);
    return Vec3(a, b, c);
  }
  Vec4 vec4() {
    expect(T_LPAREN);
    double a = scalar(); expect(T_COMMA);
    double b = scalar(); expect(T_COMMA);
    double c = scalar(); expect(T_COMMA);
    double d = scalar(); expect(T_RPAREN);
    return Vec4(a, b, c, d);
  }
  // "<keyword> = value ;" forms (Parser.cpp:904-957)
  double scalarExpression() { advance(); expect(T_EQUALS); double v = scalar(); condRead(T_SEMI); return v; }
  Vec3 vec3Expression() { advance(); expect(T_EQUALS); Vec3 v = vec3(); condRead(T_SEMI); return v; }
  Vec4 vec4Expression() { advance(); expect(T_EQUALS); Vec4 v = vec4(); condRead(T_SEMI); return v; }
  bool boolExpression() {
    advance();
    expect(T_EQUALS);
    bool v;
    if (isKeyword("true")) v = true;
    else if (isKeyword("false")) v = false;
    else lex_.fail("Expected boolean");
    advance();
    condRead(T_SEMI);
    return v;
  }
  void identExpression() {
    advance();
    expect(T_EQUALS);
    if (cur_.kind != T_IDENT) lex_.fail("IDENT expected");
    advance();
    condRead(T_SEMI);
  }

  bool isGeometryStart() const {
    if (cur_.kind == T_LBRACE) return true;
    static const char* const g[] = {"sphere", "box", "square", "cylinder", "cone", "trimesh",
                                    "translate", "rotate", "scale", "transform"};
    for (const char* k : g)
      if (isKeyword(k)) return true;
    return false;
  }

  // ---- camera (Parser.cpp:85-148)
  void camera(Scene& s) {
    bool hasView = false, hasUp = false;
    Vec3 viewDir, upDir;
    advance();
    expect(T_LBRACE);
    for (;;) {
      if (isKeyword("position")) s.camera.setEye(vec3Expression());
      else if (isKeyword("fov")) s.camera.setFOV(scalarExpression());
      else if (isKeyword("quaternian")) {
        Vec4 q = vec4Expression();
        s.camera.setLook(q[0], q[1], q[2], q[3]);
      } else if (isKeyword("aspectratio")) s.camera.setAspectRatio(scalarExpression());
      else if (isKeyword("viewdir")) { viewDir = vec3Expression(); hasView = true; }
      else if (isKeyword("updir")) { upDir = vec3Expression(); hasUp = true; }
      else if (cur_.kind == T_RBRACE) {
        if (hasView) {
          if (!hasUp) lex_.fail("Expected: 'updir'");
          s.camera.setLook(viewDir, upDir);
        } else if (hasUp) {
          lex_.fail("Expected: 'viewdir'");
        }
        advance();
        return;
      } else {
        lex_.fail("Expected: camera attribute");
      }
    }
  }

  // ---- geometry
  void transformable(Scene& s, TransformNode* xf, const Material& mat) {
    if (cur_.kind == T_LBRACE) { group(s, xf, mat); return; }
    if (isKeyword("sphere")) simplePrimitive(s, xf, mat, RT_OBJ_SPHERE);
    else if (isKeyword("box")) simplePrimitive(s, xf, mat, RT_OBJ_BOX);
    else if (isKeyword("square")) simplePrimitive(s, xf, mat, RT_OBJ_SQUARE);
    else if (isKeyword("cylinder")) simplePrimitive(s, xf, mat, RT_OBJ_CYLINDER);
    else if (isKeyword("cone")) cone(s, xf, mat);
    else if (isKeyword("trimesh")) trimesh(s, xf, mat);
    else if (isKeyword("translate")) {
      advance(); expect(T_LPAREN);
      double x = scalar(); expect(T_COMMA);
      double y = scalar(); expect(T_COMMA);
      double z = scalar(); expect(T_COMMA);
      transformable(s, xf->createChild(translation(x, y, z)), mat);
      expect(T_RPAREN); condRead(T_SEMI);
    } else if (isKeyword("rotate")) {
      advance(); expect(T_LPAREN);
      double x = scalar(); expect(T_COMMA);
      double y = scalar(); expect(T_COMMA);
      double z = scalar(); expect(T_COMMA);
      double w = scalar(); expect(T_COMMA);
      transformable(s, xf->createChild(rotation(w, (float)x, (float)y, (float)z)), mat);
      expect(T_RPAREN); condRead(T_SEMI);
    } else if (isKeyword("scale")) {
      advance(); expect(T_LPAREN);
      double x = scalar(), y, z;
      expect(T_COMMA);
      if (cur_.kind == T_SCALAR) {
        y = scalar(); expect(T_COMMA);
        z = scalar(); expect(T_COMMA);
      } else {
        y = z = x;
      }
      transformable(s, xf->createChild(scaling(x, y, z)), mat);
      expect(T_RPAREN); condRead(T_SEMI);
    } else if (isKeyword("transform")) {
      advance(); expect(T_LPAREN);
      Mat4 m;
      for (int r = 0; r < 4; r++) {
        Vec4 row = vec4();
        expect(T_COMMA);
        for (int c = 0; c < 4; c++) m[r][c] = row[c];
      }
      transformable(s, xf->createChild(m), mat);
      expect(T_RPAREN); condRead(T_SEMI);
    } else {
      lex_.fail("Expected: transformable element");
    }
  }

  // Parser.cpp:176-210. A material statement inside a group always raises a
  // syntax error in the reference (the case falls through to default), so it
  // does here as well.
  void group(Scene& s, TransformNode* xf, const Material& mat) {
    expect(T_LBRACE);
    for (;;) {
      if (cur_.kind == T_RBRACE) { advance(); return; }
      if (isGeometryStart()) transformable(s, xf, mat);
      else lex_.fail("Expected: '}' or geometry");
    }
  }

  std::unique_ptr<Geometry> newGeometry(Scene& s, int kind, const TransformNode* xf,
                                        const Material& m) {
    std::unique_ptr<Geometry> g(new Geometry);
    g->kind = kind;
    g->objectID = s.nextObjectID++;
    g->transform = xf;
    g->material = m;
    return g;
  }

  // sphere / box / square / cylinder (Parser.cpp:356-490)
  void simplePrimitive(Scene& s, TransformNode* xf, const Material& mat, int kind) {
    advance();
    expect(T_LBRACE);
    Material m = mat;
    for (;;) {
      if (isKeyword("material")) m = materialExpression(mat);
      else if (isKeyword("name")) identExpression();
      else if (cur_.kind == T_RBRACE) {
        advance();
        s.add(newGeometry(s, kind, xf, m));
        return;
      } else {
        lex_.fail("Expected: primitive attributes");
      }
    }
  }

  // Parser.cpp:492-541 and the constructor arithmetic of Cone.h:10-37
  void cone(Scene& s, TransformNode* xf, const Material& mat) {
    advance();
    expect(T_LBRACE);
    Material m = mat;
    double bottomRadius = 1.0, topRadius = 0.0, height = 1.0;
    bool capped = true;
    for (;;) {
      if (isKeyword("material")) m = materialExpression(mat);
      else if (isKeyword("name")) identExpression();
      else if (isKeyword("capped")) capped = boolExpression();
      else if (isKeyword("bottom_radius")) bottomRadius = scalarExpression();
      else if (isKeyword("top_radius")) topRadius = scalarExpression();
      else if (isKeyword("height")) height = scalarExpression();
      else if (cur_.kind == T_RBRACE) {
        advance();
        std::unique_ptr<Geometry> g = newGeometry(s, RT_OBJ_CONE, xf, m);
        g->height = height;
        g->bRadius = (bottomRadius < 0.0f) ? -bottomRadius : bottomRadius;
        g->tRadius = (topRadius < 0.0f) ? -topRadius : topRadius;
        g->capped = capped;
        if (g->bRadius < 0.0001) g->bRadius = 0.0001;
        if (g->tRadius < 0.0001) g->tRadius = 0.0001;
        double beta = (g->tRadius - g->bRadius) / g->height;
        if (std::fabs(beta) < 0.001) beta = 0.001;
        double gamma = (beta < 0.0) ? g->tRadius / beta : g->bRadius / beta;
        g->betaSquared = beta * beta;
        if (gamma < 0.0) gamma = gamma - g->height;
        g->gamma = gamma;
        s.add(std::move(g));
        return;
      } else {
        lex_.fail("Expected: cone attributes");
      }
    }
  }

  // Parser.cpp:543-706 + Trimesh::addFace / TrimeshFace ctor (trimesh.cpp:34-48, trimesh.h:90-116)
  void trimesh(Scene& s, TransformNode* xf, const Material& mat) {
    // The Trimesh object takes its ID before any of its faces do.
    std::unique_ptr<Geometry> g = newGeometry(s, RT_OBJ_TRIMESH, xf, mat);
    g->mesh.reset(new Trimesh);
    Trimesh& tm = *g->mesh;
    advance();
    expect(T_LBRACE);
    bool genNormals = false;
    std::list<Vec3> faces;
    for (;;) {
      if (isKeyword("gennormals")) {
        advance(); expect(T_SEMI);
        genNormals = true;
      } else if (isKeyword("material")) {
        g->material = materialExpression(mat);
      } else if (isKeyword("name")) {
        identExpression();
      } else if (isKeyword("materials")) {
        advance(); expect(T_EQUALS); expect(T_LPAREN);
        if (cur_.kind != T_RPAREN) {
          tm.materials.push_back(material(g->material));
          while (cur_.kind != T_RPAREN) {
            expect(T_COMMA);
            tm.materials.push_back(material(g->material));
          }
        }
        expect(T_RPAREN); expect(T_SEMI);
      } else if (isKeyword("normals")) {
        advance(); expect(T_EQUALS); expect(T_LPAREN);
        if (cur_.kind != T_RPAREN) {
          tm.normals.push_back(vec3());
          while (cur_.kind != T_RPAREN) { expect(T_COMMA); tm.normals.push_back(vec3()); }
        }
        expect(T_RPAREN); expect(T_SEMI);
      } else if (isKeyword("faces")) {
        advance(); expect(T_EQUALS); expect(T_LPAREN);
        if (cur_.kind != T_RPAREN) {
          faceList(faces);
          while (cur_.kind != T_RPAREN) { expect(T_COMMA); faceList(faces); }
        }
        expect(T_RPAREN); expect(T_SEMI);
      } else if (isKeyword("points")) {
        advance(); expect(T_EQUALS); expect(T_LPAREN);
        if (cur_.kind != T_RPAREN) {
          tm.vertices.push_back(vec3());
          while (cur_.kind != T_RPAREN) { expect(T_COMMA); tm.vertices.push_back(vec3()); }
        }
        expect(T_RPAREN); expect(T_SEMI);
      } else if (cur_.kind == T_RBRACE) {
        advance();
        for (const Vec3& f : faces) addFace(s, tm, (int)f[0], (int)f[1], (int)f[2]);
        if (genNormals) tm.generateNormals();
        if (!tm.materials.empty() && tm.materials.size() != tm.vertices.size())
          throw std::runtime_error("Bad Trimesh: Wrong number of materials.");
        if (!tm.normals.empty() && tm.normals.size() != tm.vertices.size())
          throw std::runtime_error("Bad Trimesh: Wrong number of normals.");
        // Explicit "normals=" lists never set Trimesh::vertNorms in the
        // reference (only generateNormals does, trimesh.cpp:237), so they are
        // parsed, size-checked and then ignored by shading.
        s.add(std::move(g));
        return;
      } else {
        lex_.fail("Expected: trimesh attributes");
      }
    }
  }

  void faceList(std::list<Vec3>& faces) {  // fan triangulation, Parser.cpp:688-706
    std::vector<double> pts;
    expect(T_LPAREN);
    if (cur_.kind != T_RPAREN) {
      pts.push_back(scalar());
      while (cur_.kind != T_RPAREN) { expect(T_COMMA); pts.push_back(scalar()); }
    }
    expect(T_RPAREN);
    if (pts.size() < 3) lex_.fail("Faces must have at least 3 vertices.");
    double a = pts[0], b = pts[1];
    for (size_t i = 2; i < pts.size(); i++) {
      faces.push_back(Vec3(a, b, pts[i]));
      b = pts[i];
    }
  }

  void addFace(Scene& s, Trimesh& tm, int a, int b, int c) {
    int vcnt = (int)tm.vertices.size();
    if (a >= vcnt || b >= vcnt || c >= vcnt || a < 0 || b < 0 || c < 0) {
      std::ostringstream os;
      os << "Bad face in trimesh: (" << a << ", " << b << ", " << c << ")";
      throw std::runtime_error(os.str());
    }
    TrimeshFace f;
    f.ids[0] = a; f.ids[1] = b; f.ids[2] = c;
    f.objectID = s.nextObjectID++;       // degenerate faces consume an ID too
    const Vec3 &pa = tm.vertices[a], &pb = tm.vertices[b], &pc = tm.vertices[c];
    Vec3 vab = sub(pb, pa), vac = sub(pc, pa), vcb = sub(pb, pc);
    if (iszero(vab) || iszero(vac) || iszero(vcb)) return;  // degen: dropped
    f.normal = normalized(cross(sub(pb, pa), sub(pc, pa)));
    Vec3 hi = vmax(pa, pb), lo = vmin(pa, pb);
    hi = vmax(pc, hi);
    lo = vmin(pc, lo);
    f.bounds = BoundingBox(lo, hi);
    tm.faces.push_back(f);
  }

  // ---- lights (Parser.cpp:724-900)
  void pointLight(Scene& s) {
    Light l;
    l.type = RT_LIGHT_POINT;
    bool hasPos = false, hasColor = false;
    advance(); expect(T_LBRACE);
    for (;;) {
      if (isKeyword("position")) {
        if (hasPos) lex_.fail("Repeated 'position' attribute");
        l.position = vec3Expression(); hasPos = true;
      } else if (isKeyword("color")) {
        if (hasColor) lex_.fail("Repeated 'color' attribute");
        l.color = vec3Expression(); hasColor = true;
      } else if (isKeyword("constant_attenuation_coeff")) l.constantTerm = (float)scalarExpression();
      else if (isKeyword("linear_attenuation_coeff")) l.linearTerm = (float)scalarExpression();
      else if (isKeyword("quadratic_attenuation_coeff")) l.quadraticTerm = (float)scalarExpression();
      else if (cur_.kind == T_RBRACE) {
        if (!hasColor) lex_.fail("Expected: 'color'");
        if (!hasPos) lex_.fail("Expected: 'position'");
        advance();
        s.lights.push_back(l);
        return;
      } else {
        lex_.fail("expecting 'position' or 'color' attribute, or an attenuation coefficient");
      }
    }
  }

  void spotLight(Scene& s) {
    Light l;
    l.type = RT_LIGHT_SPOT;
    bool hasDir = false, hasColor = false, hasPos = false, hasAngle = false, hasFall = false;
    advance(); expect(T_LBRACE);
    for (;;) {
      if (isKeyword("direction")) {
        if (hasDir) lex_.fail("Repeated 'direction' attribute");
        l.orientation = vec3Expression(); hasDir = true;
      } else if (isKeyword("position")) {
        if (hasPos) lex_.fail("Repeated 'position' attribute");
        l.position = vec3Expression(); hasPos = true;
      } else if (isKeyword("atten_angle")) {
        if (hasAngle) lex_.fail("Repeated 'angle' attribute");
        l.attenAngle = scalarExpression(); hasAngle = true;
      } else if (isKeyword("fall_rate")) {
        if (hasFall) lex_.fail("Repeated 'fall_rate' attribute");
        l.fallRate = scalarExpression(); hasFall = true;
      } else if (isKeyword("color")) {
        if (hasColor) lex_.fail("Repeated 'color' attribute");
        l.color = vec3Expression(); hasColor = true;
      } else if (cur_.kind == T_RBRACE) {
        if (!hasColor) lex_.fail("Expected: 'color'");
        if (!hasDir) lex_.fail("Expected: 'direction'");
        if (!hasAngle) lex_.fail("Expected: 'angle'");
        if (!hasPos) lex_.fail("Expected: 'position'");
        if (!hasFall) lex_.fail("Expected: 'fall_rate'");
        advance();
        l.orientation = normalized(l.orientation);  // SpotLight ctor, light.h:56
        s.lights.push_back(l);
        return;
      } else {
        lex_.fail("expecting 'position' or 'color' attribute");
      }
    }
  }

  void directionalLight(Scene& s) {
    Light l;
    l.type = RT_LIGHT_DIRECTIONAL;
    bool hasDir = false, hasColor = false;
    advance(); expect(T_LBRACE);
    for (;;) {
      if (isKeyword("direction")) {
        if (hasDir) lex_.fail("Repeated 'direction' attribute");
        l.orientation = vec3Expression(); hasDir = true;
      } else if (isKeyword("color")) {
        if (hasColor) lex_.fail("Repeated 'color' attribute");
        l.color = vec3Expression(); hasColor = true;
      } else if (cur_.kind == T_RBRACE) {
        if (!hasColor) lex_.fail("Expected: 'color'");
        if (!hasDir) lex_.fail("Expected: 'position'");
        advance();
        l.orientation = normalized(l.orientation);  // DirectionalLight ctor, light.h:37
        s.lights.push_back(l);
        return;
      } else {
        lex_.fail("expecting 'position' or 'color' attribute");
      }
    }
  }

  // ---- materials (Parser.cpp:941-948,1046-1172)
  Material materialExpression(const Material& parent) {
    expectKeyword("material");
    expect(T_EQUALS);
    Material m = material(parent);
    condRead(T_SEMI);
    return m;
  }

  Material material(const Material& parent) {
    if (cur_.kind == T_IDENT)
      // The reference looks the name up without consuming the token, so every
      // use of a named material ends in a syntax error one token later.
      lex_.fail("named material references are not usable in this scene language");
    expect(T_LBRACE);
    Material m = parent;
    std::string name;
    for (;;) {
      if (isKeyword("emissive")) m.ke = vec3Parameter();
      else if (isKeyword("ambient")) m.ka = vec3Parameter();
      else if (isKeyword("specular")) m.ks = vec3Parameter();
      else if (isKeyword("diffuse")) m.kd = vec3Parameter();
      else if (isKeyword("reflective")) m.kr = vec3Parameter();
      else if (isKeyword("transmissive")) m.kt = vec3Parameter();
      else if (isKeyword("index")) m.index = scalarParameter();
      else if (isKeyword("shininess")) m.shininess = scalarParameter();
      else if (isKeyword("name")) {
        advance();
        if (cur_.kind != T_IDENT) lex_.fail("IDENT expected");
        name = cur_.text;
        advance();
        expect(T_SEMI);
      } else if (cur_.kind == T_RBRACE) {
        advance();
        if (!name.empty()) {
          if (named_.count(name)) lex_.fail("Redefinition of material '" + name + "'.");
          named_[name] = m;
        }
        return m;
      } else {
        lex_.fail("Expected: material attribute");
      }
    }
  }

  MaterialParameter vec3Parameter() {
    MaterialParameter p;
    advance(); expect(T_EQUALS);
    if (isKeyword("map")) {
      advance(); expect(T_LPAREN);
      if (cur_.kind != T_IDENT) lex_.fail("IDENT expected");
      advance(); expect(T_RPAREN); condRead(T_SEMI);
      p.mapped = true;
    } else {
      p.value = vec3();
      condRead(T_SEMI);
    }
    return p;
  }

  MaterialParameter scalarParameter() {
    MaterialParameter p;
    advance(); expect(T_EQUALS);
    if (isKeyword("map")) {
      advance(); expect(T_LPAREN);
      if (cur_.kind != T_IDENT) lex_.fail("IDENT expected");
      advance(); expect(T_RPAREN); condRead(T_SEMI);
      p.mapped = true;
    } else {
      double v = scalar();
      condRead(T_SEMI);
      p.value = Vec3(v, v, v);
    }
    return p;
  }

  Lexer lex_;
  Token cur_;
  std::map<std::string, Material> named_;
};

}  // namespace

std::unique_ptr<Scene> parseSceneText(const std::string& text) {
  Parser p(text);
  std::unique_ptr<Scene> s = p.parseScene();
  for (const auto& g : s->objects)
    if (g->kind == RT_OBJ_TRIMESH)
      for (const Material& m : g->mesh->materials)
        if (m.ke.mapped || m.ka.mapped || m.ks.mapped || m.kd.mapped || m.kr.mapped ||
            m.kt.mapped || m.shininess.mapped || m.index.mapped)
          throw std::runtime_error("texture-mapped parameters in per-vertex materials are not supported");
  return s;
}

// RayTracer::loadScene, src/RayTracer.cpp:268-321
std::unique_ptr<Scene> loadSceneFile(const std::string& path, const SceneOptions& opt,
                                     std::string* error) {
  std::ifstream ifs(path.c_str(), std::ios::binary);
  if (!ifs) {
    if (error) *error = "Error: couldn't read scene file " + path;
    return nullptr;
  }
  std::ostringstream ss;
  ss << ifs.rdbuf();
  try {
    std::unique_ptr<Scene> s = parseSceneText(ss.str());
    applyOptions(*s, opt, path);
    return s;
  } catch (const std::exception& e) {
    if (error) *error = e.what();
    return nullptr;
  }
}

}  // namespace rta
