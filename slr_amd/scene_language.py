"""Loader for the asset-free subset of the reference's scene language.

The reference describes scenes in a small scripting language (grammar: libSLRSceneGraph/Parser/SceneParser.yy;
built-in functions: libSLRSceneGraph/API.cpp:96-1110) and flattens the node graph before render().  This module is
an independent interpreter for that language — tokenizer, recursive-descent parser for the grammar, tree-walking
evaluator with the reference's argument binding (named parameters first, then every positional value goes to the first
unassigned parameter of a convertible type; overloads are tried in order: SceneParser.cpp:399-453) — whose built-ins
construct a flat `slr_amd.abi.Scene` through `scenes.SceneBuilder` instead of libSLR objects.

What loads: createMesh / createVertex geometry, node hierarchy with static transforms, constant Spectrum / texture
values, matte / metal / glass / microfacet / Ward / Ashikhmin materials, diffuse emitters, the perspective camera, setRenderer,
setRenderSettings — i.e. the walls, lights, cameras and materials of TestScenes/Cornell_Box_*.txt.  What the image
lacks (no assimp / OpenEXR, no asset files in the reference tree): load3DModel accepts only the two primitive models
the test scenes use ("…/sphere.assbin", "…/box.assbin") and substitutes this package's tessellated unit sphere / cube;
Image2D textures, setEnvironment images, "sum" / "mix" materials nested more than one level deep and the scan* helpers raise UnsupportedFeature
("sum" / "mix" of two single-lobe materials, either possibly "inverse", load as SLRHIP_MATERIAL_MULTI).
"""
import math
import re

import numpy as np

from . import abi, scenes, spectra


class SceneLanguageError(RuntimeError):
    pass


class UnsupportedFeature(SceneLanguageError):
    pass


# ---------------------------------------------------------------------------------------------- tokenizer
_TOKEN = re.compile(r"""
    (?P<ws>\s+|//[^\n]*|/\*.*?\*/)
  | (?P<real>(\d+\.\d*|\.\d+)([eE][+-]?\d+)?|\d+[eE][+-]?\d+)
  | (?P<int>\d+)
  | (?P<str>"(?:[^"\\]|\\.)*")
  | (?P<id>[A-Za-z_][A-Za-z_0-9]*)
  | (?P<op>\+\+|--|\+=|-=|\*=|/=|%=|==|!=|<=|>=|&&|\|\||[-+*/%<>=!(){}\[\],;:])
""", re.X | re.S)
_KEYWORDS = {"if", "else", "for", "function", "return", "true", "false"}


def tokenize(text):
    out, pos = [], 0
    while pos < len(text):
        m = _TOKEN.match(text, pos)
        if not m:
            raise SceneLanguageError("unexpected character %r at offset %d" % (text[pos], pos))
        pos = m.end()
        kind = m.lastgroup
        if kind == "ws":
            continue
        val = m.group(kind)
        if kind == "id" and val in _KEYWORDS:
            kind = val
        out.append((kind, val))
    out.append(("eof", ""))
    return out


# ---------------------------------------------------------------------------------------------- parser (SceneParser.yy)
class Parser:
    def __init__(self, tokens):
        self.t, self.i = tokens, 0

    def peek(self, k=0):
        return self.t[self.i + k]

    def take(self, kind=None, val=None):
        tok = self.t[self.i]
        if (kind and tok[0] != kind) or (val is not None and tok[1] != val):
            raise SceneLanguageError("expected %s %s, found %r" % (kind or "", val or "", tok))
        self.i += 1
        return tok

    def at(self, val):
        return self.t[self.i][0] == "op" and self.t[self.i][1] == val

    def statements(self, until=None):
        out = []
        while self.peek()[0] != "eof" and not (until and self.at(until)):
            out.append(self.statement())
        return out

    def statement(self):
        k, v = self.peek()
        if self.at("{"):
            self.take()
            body = self.statements("}")
            self.take("op", "}")
            return ("block", body)
        if k == "if":
            self.take(); self.take("op", "(")
            cond = self.expression(); self.take("op", ")")
            then = self.statement()
            other = None
            if self.peek()[0] == "else":
                self.take(); other = self.statement()
            return ("if", cond, then, other)
        if k == "for":
            self.take(); self.take("op", "(")
            init = self.expression(); self.take("op", ";")
            cond = self.expression(); self.take("op", ";")
            step = self.expression(); self.take("op", ")")
            return ("for", init, cond, step, self.statement())
        if k == "function":
            self.take()
            name = self.take("id")[1]
            self.take("op", "(")
            params = []
            while not self.at(")"):
                pname = self.take("id")[1]
                default = None
                if self.at("="):
                    self.take(); default = self.expression()
                params.append((pname, default))
                if self.at(","):
                    self.take()
            self.take("op", ")")
            return ("function", name, params, self.statement())
        if k == "return":
            self.take()
            if self.at(";"):
                self.take(); return ("return", None)
            e = self.expression(); self.take("op", ";")
            return ("return", e)
        e = self.expression()
        self.take("op", ";")
        return ("expr", e)

    # precedence of the %left / %right declarations: = += … < || < && < == != < < > <= >= < + - < * / %
    def expression(self):
        if self.peek()[0] == "id" and self.peek(1)[0] == "op" and self.peek(1)[1] in ("=", "+=", "-=", "*=", "/=", "%="):
            name = self.take()[1]
            op = self.take()[1]
            return ("assign", name, op, self.expression())
        return self.binary(0)

    _LEVELS = [("||",), ("&&",), ("==", "!="), ("<", ">", "<=", ">="), ("+", "-"), ("*", "/", "%")]

    def binary(self, level):
        if level == len(self._LEVELS):
            return self.unary()
        left = self.binary(level + 1)
        while self.peek()[0] == "op" and self.peek()[1] in self._LEVELS[level]:
            op = self.take()[1]
            left = ("bin", op, left, self.binary(level + 1))
        return left

    def unary(self):
        if self.peek()[0] == "op" and self.peek()[1] in ("+", "-", "!"):
            op = self.take()[1]
            return ("un", op, self.unary())
        if self.peek()[0] == "op" and self.peek()[1] in ("++", "--"):
            op = self.take()[1]
            return ("preinc", op, self.take("id")[1])
        if self.peek()[0] == "id" and self.peek(1)[0] == "op" and self.peek(1)[1] in ("++", "--"):
            name = self.take()[1]
            return ("postinc", self.take()[1], name)
        return self.postfix(self.single())

    def postfix(self, e):
        while self.at("["):
            self.take()
            idx = self.expression()
            self.take("op", "]")
            e = ("index", e, idx)
        return e

    def parameter(self):
        e = self.expression()
        if self.at(":"):
            self.take()
            return (e, self.expression())
        return (None, e)

    def single(self):
        k, v = self.peek()
        if k == "real":
            self.take(); return ("const", float(v))
        if k == "int":
            self.take(); return ("const", int(v))
        if k == "str":
            self.take(); return ("const", bytes(v[1:-1], "utf-8").decode("unicode_escape"))
        if k in ("true", "false"):
            self.take(); return ("const", k == "true")
        if k == "id":
            self.take()
            if self.at("("):
                self.take()
                args = []
                while not self.at(")"):
                    args.append(self.parameter())
                    if self.at(","):
                        self.take()
                self.take("op", ")")
                return ("call", v, args)
            return ("var", v)
        if self.at("("):
            self.take()
            if self.at(","):                                  # "(" "," ")": the empty tuple
                self.take(); self.take("op", ")")
                return ("tuple", [])
            first = self.parameter()
            if self.at(")"):
                self.take()
                if first[0] is None:
                    return first[1]                           # "(" Expression ")"
                return ("tuple", [first])
            elems = [first]
            while self.at(","):
                self.take()
                if self.at(")"):
                    break                                     # "(" Parameter "," ")"
                elems.append(self.parameter())
            self.take("op", ")")
            return ("tuple", elems)
        raise SceneLanguageError("unexpected token %r" % (self.peek(),))


# ---------------------------------------------------------------------------------------------- values
class Tuple:
    """ParameterList: ordered (key or None, value) pairs."""

    def __init__(self, items=()):
        self.items = list(items)

    def unnamed(self):
        return [v for k, v in self.items if k is None]

    def __len__(self):
        return len(self.items)

    def index(self, key):
        if isinstance(key, str):
            for k, v in self.items:
                if k == key:
                    return v
            return Tuple()          # attrs["missing"] -> an empty tuple (numElements == 0)
        return self.unnamed()[int(key)]


class Vec:
    def __init__(self, x, y, z):
        self.v = np.array([x, y, z], dtype=np.float64)


class SpectrumValue:
    """A constant spectrum before it is bound to a SceneBuilder: constructor name + arguments, and a scale factor
    (`Spectrum * x` / `x * Spectrum`, API.cpp:443-462)."""

    def __init__(self, ctor, args, scale=1.0):
        self.ctor, self.args, self.scale = ctor, args, scale


class SpectrumTex:
    def __init__(self, spectrum):
        self.spectrum = spectrum


class NormalTex:
    pass


class FloatTex:
    def __init__(self, value):
        self.value = float(value)


class Material:
    def __init__(self, kind, params, emitter=None):
        self.kind, self.params, self.emitter = kind, params, emitter


class Emitter:
    def __init__(self, emittance):
        self.emittance = emittance


class Mesh:
    def __init__(self, vertices, groups):
        self.vertices, self.groups = vertices, groups


class Model:
    def __init__(self, kind, material):
        self.kind, self.material = kind, material


class Camera:
    def __init__(self, **kw):
        self.__dict__.update(kw)


class Node:
    def __init__(self):
        self.transform = np.eye(4)
        self.children = []


class ReferenceNode:
    """createReferenceNode (API.cpp:745-752): a second placement of a node's subtree.  The reference turns it into a
    TransformedSurfaceObject over the subtree's aggregate (libSLRSceneGraph/nodes.cpp:174-184); here it becomes a slrhip_instance."""
    def __init__(self, node):
        self.node = node


class Function:
    def __init__(self, overloads):
        self.overloads = overloads          # list of (signature, callable(args dict) or AST body)


NUM, STR, TUP, ANY, FUNC = "num", "str", "tuple", "any", "func"


def _convertible(value, expected):
    if expected == ANY:
        return True
    if expected == NUM:
        return isinstance(value, (int, float)) and not isinstance(value, bool) or isinstance(value, bool)
    if expected == STR:
        return isinstance(value, str)
    if expected == TUP:
        return isinstance(value, Tuple)
    if expected == FUNC:
        return isinstance(value, Function) or value is None
    return isinstance(value, expected)


_NODEFAULT = object()


def bind(signature, params):
    """mapParamsToArgs, SceneParser.cpp:399-453.  signature: [(name, type, default or _NODEFAULT)]."""
    args, assigned = {}, [False] * len(signature)
    for key, value in params.items:
        if key is None:
            continue
        for i, (name, typ, _) in enumerate(signature):
            if name == key and _convertible(value, typ):
                args[name], assigned[i] = value, True
                break
        else:
            return None
    for key, value in params.items:
        if key is not None:
            continue
        for i, (name, typ, _) in enumerate(signature):
            if not assigned[i] and _convertible(value, typ):
                args[name], assigned[i] = value, True
                break
        else:
            return None
    for i, (name, _, default) in enumerate(signature):
        if not assigned[i]:
            if default is _NODEFAULT:
                return None
            args[name] = default
    return args


class _Return(Exception):
    def __init__(self, value):
        self.value = value


# ---------------------------------------------------------------------------------------------- interpreter
class Interpreter:
    def __init__(self, sphere_tessellation=(48, 24)):
        self.root = Node()
        self.globals = {"root": self.root}
        self.scopes = [self.globals]
        self.renderer = {"method": "PT", "samples": 8}
        self.settings = {"width": 1024, "height": 1024, "timeStart": 0.0, "timeEnd": 0.0, "brightness": 1.0, "rngSeed": 1509761209}
        self.sphere_tessellation = sphere_tessellation
        self._install_builtins()

    # ---- evaluation -------------------------------------------------------------------------------------
    def run(self, text):
        for st in Parser(tokenize(text)).statements():
            self.exec(st)

    def lookup(self, name):
        for sc in reversed(self.scopes):
            if name in sc:
                return sc[name]
        raise SceneLanguageError("undefined variable %s" % name)

    def assign(self, name, value):
        for sc in reversed(self.scopes):
            if name in sc:
                sc[name] = value
                return
        self.scopes[-1][name] = value

    def exec(self, st):
        kind = st[0]
        if kind == "expr":
            self.eval(st[1])
        elif kind == "block":
            for s in st[1]:
                self.exec(s)
        elif kind == "if":
            if self.eval(st[1]):
                self.exec(st[2])
            elif st[3] is not None:
                self.exec(st[3])
        elif kind == "for":
            self.eval(st[1])
            while self.eval(st[2]):
                self.exec(st[4])
                self.eval(st[3])
        elif kind == "function":
            _, name, params, body = st
            sig = [(p, ANY, _NODEFAULT if d is None else self.eval(d)) for p, d in params]
            self.assign(name, Function([(sig, body)]))
        elif kind == "return":
            raise _Return(None if st[1] is None else self.eval(st[1]))

    def call(self, fn, params):
        if not isinstance(fn, Function):
            raise SceneLanguageError("call of a non-function value")
        for sig, impl in fn.overloads:
            args = bind(sig, params)
            if args is None:
                continue
            if callable(impl):
                return impl(args)
            self.scopes.append(dict(args))
            try:
                self.exec(impl)
                return None
            except _Return as r:
                return r.value
            finally:
                self.scopes.pop()
        raise SceneLanguageError("Parameters are invalid.")

    def eval(self, e):
        kind = e[0]
        if kind == "const":
            return e[1]
        if kind == "var":
            return self.lookup(e[1])
        if kind == "tuple":
            return Tuple([(None if k is None else self.eval(k), self.eval(v)) for k, v in e[1]])
        if kind == "call":
            params = Tuple([(None if k is None else self.eval(k), self.eval(v)) for k, v in e[2]])
            try:
                return self.call(self.lookup(e[1]), params)
            except SceneLanguageError as err:
                if str(err) == "Parameters are invalid.":
                    raise SceneLanguageError("%s(...): parameters are invalid" % e[1]) from None
                raise
        if kind == "index":
            base = self.eval(e[1])
            if not isinstance(base, Tuple):
                raise SceneLanguageError("[] applied to a non-tuple")
            return base.index(self.eval(e[2]))
        if kind == "assign":
            _, name, op, rhs = e
            value = self.eval(rhs)
            if op != "=":
                value = self.binop(op[0], self.lookup(name), value)
            self.assign(name, value)
            return value
        if kind == "un":
            v = self.eval(e[2])
            if e[1] == "!":
                return not v
            if isinstance(v, Vec):
                return Vec(*(-v.v if e[1] == "-" else v.v))
            return -v if e[1] == "-" else v
        if kind in ("preinc", "postinc"):
            old = self.lookup(e[2])
            new = old + (1 if e[1] == "++" else -1)
            self.assign(e[2], new)
            return new if kind == "preinc" else old
        if kind == "bin":
            if e[1] == "&&":
                return bool(self.eval(e[2])) and bool(self.eval(e[3]))
            if e[1] == "||":
                return bool(self.eval(e[2])) or bool(self.eval(e[3]))
            return self.binop(e[1], self.eval(e[2]), self.eval(e[3]))
        raise SceneLanguageError("cannot evaluate %r" % (kind,))

    @staticmethod
    def binop(op, a, b):
        if op == "*" and isinstance(a, np.ndarray) and isinstance(b, np.ndarray):
            return a @ b                                               # Matrix4x4 product
        if op in ("*", "/") and isinstance(a, SpectrumValue) and isinstance(b, (int, float)):
            return SpectrumValue(a.ctor, a.args, a.scale * b if op == "*" else a.scale / b)
        if op == "*" and isinstance(b, SpectrumValue) and isinstance(a, (int, float)):
            return SpectrumValue(b.ctor, b.args, b.scale * a)
        if isinstance(a, Vec) or isinstance(b, Vec):
            av = a.v if isinstance(a, Vec) else a
            bv = b.v if isinstance(b, Vec) else b
            r = {"+": lambda: av + bv, "-": lambda: av - bv, "*": lambda: av * bv, "/": lambda: av / bv}[op]()
            return Vec(*r)
        if op == "+" and isinstance(a, str):
            return a + str(b)
        ops = {"+": lambda: a + b, "-": lambda: a - b, "*": lambda: a * b, "%": lambda: a % b,
               "<": lambda: a < b, ">": lambda: a > b, "<=": lambda: a <= b, ">=": lambda: a >= b,
               "==": lambda: a == b, "!=": lambda: a != b}
        if op == "/":
            if isinstance(a, int) and isinstance(b, int) and not isinstance(a, bool):
                return int(a / b)                                      # C++ integer division
            return a / b
        return ops[op]()

    # ---- built-ins (libSLRSceneGraph/API.cpp) ------------------------------------------------------------
    def _def(self, name, *overloads):
        self.globals[name] = Function(list(overloads))

    def _install_builtins(self):
        D = _NODEFAULT
        f = self._def
        f("print", ([("value", ANY, D)], lambda a: print(a["value"])))
        f("numElements", ([("tuple", TUP, D)], lambda a: len(a["tuple"])))
        f("addItem", ([("tuple", TUP, D), ("key", STR, ""), ("item", ANY, D)],
                      lambda a: Tuple(a["tuple"].items + [(a["key"] or None, a["item"])])))
        f("Vector", ([("x", NUM, D), ("y", NUM, D), ("z", NUM, D)], lambda a: Vec(a["x"], a["y"], a["z"])))
        f("Point", ([("x", NUM, D), ("y", NUM, D), ("z", NUM, D)], lambda a: Vec(a["x"], a["y"], a["z"])))
        for i, n in enumerate(("getX", "getY", "getZ")):
            f(n, ([("v", Vec, D)], lambda a, i=i: float(a["v"].v[i])))
        for n in ("sqrt", "sin", "cos", "tan", "asin", "acos", "atan"):
            f(n, ([("x", NUM, D)], lambda a, fn=getattr(math, n): fn(a["x"])))
        f("pow", ([("x", NUM, D), ("e", NUM, D)], lambda a: math.pow(a["x"], a["e"])))
        f("min", ([("x0", NUM, D), ("x1", NUM, D)], lambda a: min(a["x0"], a["x1"])))
        f("max", ([("x0", NUM, D), ("x1", NUM, D)], lambda a: max(a["x0"], a["x1"])))
        f("clamp", ([("x", NUM, D), ("min", NUM, D), ("max", NUM, D)], lambda a: min(max(a["x"], a["min"]), a["max"])))
        f("dot", ([("v0", Vec, D), ("v1", Vec, D)], lambda a: float(a["v0"].v @ a["v1"].v)))
        f("cross", ([("v0", Vec, D), ("v1", Vec, D)], lambda a: Vec(*np.cross(a["v0"].v, a["v1"].v))))
        # transforms: BuiltinFunctions::Transform, Matrix4x4.cpp:97-135
        f("translate", ([("x", NUM, D), ("y", NUM, D), ("z", NUM, D)], lambda a: scenes._translate(a["x"], a["y"], a["z"])),
          ([("v", Vec, D)], lambda a: scenes._translate(*a["v"].v)))
        f("scale", ([("x", NUM, D), ("y", NUM, D), ("z", NUM, D)], lambda a: np.diag([a["x"], a["y"], a["z"], 1.0])),
          ([("s", NUM, D)], lambda a: scenes._scale(a["s"])))
        f("rotate", ([("angle", NUM, D), ("axis", Vec, D)], lambda a: scenes._rotate(a["angle"], a["axis"].v)),
          ([("angle", NUM, D), ("x", NUM, D), ("y", NUM, D), ("z", NUM, D)], lambda a: scenes._rotate(a["angle"], (a["x"], a["y"], a["z"]))))
        f("rotateX", ([("angle", NUM, D)], lambda a: scenes._rotate(a["angle"], (1, 0, 0))))
        f("rotateY", ([("angle", NUM, D)], lambda a: scenes._rotate(a["angle"], (0, 1, 0))))
        f("rotateZ", ([("angle", NUM, D)], lambda a: scenes._rotate(a["angle"], (0, 0, 1))))
        # spectra and textures: API.cpp:286-441, BuiltinFunctions::Texture
        f("Spectrum",
          ([("type", STR, D), ("value", NUM, D)], lambda a: SpectrumValue("grey", (a["type"], float(a["value"])))),
          ([("type", STR, "Reflectance"), ("space", STR, "sRGB"), ("e0", NUM, D), ("e1", NUM, D), ("e2", NUM, D)],
           lambda a: SpectrumValue("tristimulus", (a["type"], a["space"], float(a["e0"]), float(a["e1"]), float(a["e2"])))),
          ([("type", STR, "Reflectance"), ("minWL", NUM, D), ("maxWL", NUM, D), ("values", TUP, D)],
           lambda a: SpectrumValue("regular", (a["type"], float(a["minWL"]), float(a["maxWL"]), [float(v) for v in a["values"].unnamed()]))),
          ([("type", STR, "Reflectance"), ("wls", TUP, D), ("values", TUP, D)],
           lambda a: SpectrumValue("irregular", (a["type"], [float(v) for v in a["wls"].unnamed()], [float(v) for v in a["values"].unnamed()]))),
          ([("ID", STR, D), ("idx", NUM, 0)], lambda a: SpectrumValue("library", (a["ID"], int(a["idx"])))))
        f("SpectrumTexture", ([("spectrum", SpectrumValue, D)], lambda a: SpectrumTex(a["spectrum"])),
          ([("image", ANY, D)], self._unsupported("image textures (no OpenEXR / libpng in this image)")))
        f("FloatTexture", ([("value", NUM, D)], lambda a: FloatTex(a["value"])))
        f("Image2D", ([("path", STR, D), ("mode", STR, "AsIs"), ("type", STR, "Reflectance")], self._unsupported("Image2D (no image loaders, no asset files)")))
        # materials: API.cpp:472-661
        f("createSurfaceMaterial", ([("type", STR, D), ("params", TUP, D)], self._create_material))
        f("createEmitterSurfaceProperty", ([("type", STR, D), ("params", TUP, D)], self._create_emitter))
        # geometry and nodes: API.cpp:259-284,663-800
        f("createVertex", ([("position", TUP, D), ("normal", TUP, D), ("tangent", TUP, D), ("texCoord", TUP, D)],
                           lambda a: Tuple([(None, a["position"]), (None, a["normal"]), (None, a["tangent"]), (None, a["texCoord"])])))
        f("createMesh", ([("vertices", TUP, D), ("matGroups", TUP, D)], self._create_mesh))
        f("createNode", ([], lambda a: Node()))
        f("createReferenceNode", ([("node", Node, D)], lambda a: ReferenceNode(a["node"])))
        f("setTransform", ([("node", Node, D), ("transform", np.ndarray, D)], self._set_transform))
        f("addChild", ([("parent", Node, D), ("child", ANY, D)], self._add_child))
        f("load3DModel", ([("path", STR, D), ("matProc", FUNC, None)], self._load_model))
        f("createPerspectiveCamera",
          ([("sensitivity", NUM, 0.0), ("aspect", NUM, 1.0), ("fovY", NUM, 0.5235987756), ("radius", NUM, 0.0), ("imgDist", NUM, 0.02),
            ("objDist", NUM, 5.0)], lambda a: Camera(**{k: float(v) for k, v in a.items()})))
        f("setRenderer", ([("method", STR, D), ("config", TUP, Tuple())], self._set_renderer))
        f("setRenderSettings",
          ([("width", NUM, 1024), ("height", NUM, 1024), ("timeStart", NUM, 0.0), ("timeEnd", NUM, 0.0), ("brightness", NUM, 1.0),
            ("rngSeed", NUM, 1509761209)], lambda a: self.settings.update(a)))
        f("setEnvironment", ([("path", STR, D), ("scale", NUM, 1.0)], self._unsupported("setEnvironment from an image file (pass `environment=` to load_scene)")))
        for name in ("scanXZFromYPlus", "random", "lookAt", "AnimatedTransform", "NormalTexture", "Texture2DMapping", "Texture3DMapping", "copyNode"):
            f(name, ([("args", ANY, None)], self._unsupported(name)))

    @staticmethod
    def _unsupported(what):
        def raiser(_):
            raise UnsupportedFeature("not available in this build: " + what)
        return raiser

    def _set_transform(self, a):
        a["node"].transform = np.array(a["transform"], dtype=np.float64)

    def _add_child(self, a):
        a["parent"].children.append(a["child"])

    def _set_renderer(self, a):
        method = a["method"]        # recorded as written; only "PT" (the unidirectional path tracer) exists in this build,
        samples = 8                 # so a caller that renders a "BPT" scene does so with PT and should say so
        for k, v in a["config"].items:
            if k == "samples":
                samples = int(v)
        self.renderer = {"method": method, "samples": samples}

    def _create_emitter(self, a):
        if a["type"] != "diffuse":
            raise UnsupportedFeature("emitter type %r" % a["type"])
        args = bind([("emittance", SpectrumTex, _NODEFAULT)], a["params"])
        if args is None:
            raise SceneLanguageError("Parameters are invalid.")
        return Emitter(args["emittance"].spectrum)

    def _create_material(self, a):
        D = _NODEFAULT
        sigs = {
            "matte": [("reflectance", SpectrumTex, D), ("sigma", FloatTex, None)],
            "metal": [("coeffR", SpectrumTex, D), ("eta", SpectrumTex, D), ("k", SpectrumTex, D)],
            "glass": [("coeff", SpectrumTex, D), ("etaExt", SpectrumTex, D), ("etaInt", SpectrumTex, D)],
            "microfacet metal": [("eta", SpectrumTex, D), ("k", SpectrumTex, D), ("alpha_g", FloatTex, D)],
            "microfacet glass": [("etaExt", SpectrumTex, D), ("etaInt", SpectrumTex, D), ("alpha_g", FloatTex, D)],
            "Ward": [("R", SpectrumTex, D), ("anisoX", FloatTex, D), ("anisoY", FloatTex, D)],
            "Ashikhmin": [("Rd", SpectrumTex, D), ("Rs", SpectrumTex, D), ("nx", FloatTex, D), ("ny", FloatTex, D)],
            "emitter": [("scatter", Material, D), ("emitter", Emitter, D)],
            # API.cpp:583-636: InverseSurfaceMaterial / MixedSurfaceMaterial / SummedSurfaceMaterial -> SLRHIP_MATERIAL_MULTI
            "inverse": [("base", Material, D)],
            "mix": [("mat0", Material, D), ("mat1", Material, D), ("factor", FloatTex, D)],
            "sum": [("mat0", Material, D), ("mat1", Material, D)],
        }
        kind = a["type"]
        if kind not in sigs:
            raise UnsupportedFeature("surface material %r" % kind)
        args = bind(sigs[kind], a["params"])
        if args is None:
            raise SceneLanguageError("Parameters are invalid.")
        if kind == "emitter":
            base = args["scatter"]
            return Material(base.kind, base.params, args["emitter"])
        return Material(kind, args)

    def _create_mesh(self, a):
        verts = []
        for v in a["vertices"].unnamed():
            p, n, t, uv = [[float(x) for x in part.unnamed()] for part in v.unnamed()]
            verts.append((p, n, t, uv))
        groups = []
        for g in a["matGroups"].unnamed():
            args = bind([("mat", Material, _NODEFAULT), ("normal", NormalTex, None), ("alpha", FloatTex, None), ("triangles", TUP, _NODEFAULT)], g)
            if args is None:
                raise SceneLanguageError("Parameters are invalid.")
            if args["normal"] is not None or args["alpha"] is not None:
                raise UnsupportedFeature("normal / alpha textures on a mesh")
            tris = [[int(i) for i in t.unnamed()] for t in args["triangles"].unnamed()]
            groups.append((args["mat"], tris))
        return Mesh(verts, groups)

    def _load_model(self, a):
        path = a["path"].replace("\\", "/")
        base = path.rsplit("/", 1)[-1]
        if base not in ("sphere.assbin", "box.assbin"):
            raise UnsupportedFeature("load3DModel(%r): no assimp and no model files; only the sphere / box primitives are substituted" % path)
        mat = None
        if a["matProc"] is not None:
            mat = self.call(a["matProc"], Tuple([(None, base), (None, Tuple())]))
        if not isinstance(mat, Material):
            raise SceneLanguageError("load3DModel: the material procedure did not return a surface material")
        node = Node()
        node.children.append(Model("sphere" if base.startswith("sphere") else "box", mat))
        return node

    # ---- flattening (what libSLRSceneGraph does before render(): TriangleMeshNode.cpp:68-112) ------------
    def build(self, environment=None):
        b = scenes.SceneBuilder()
        spectrum_cache, material_cache = {}, {}

        def spectrum(sv):
            key = (sv.ctor, repr(sv.args), sv.scale)
            if key not in spectrum_cache:
                spectrum_cache[key] = _bind_spectrum(b, sv)
            return spectrum_cache[key]

        def material(m, see_emitter=True, multi_levels=2):
            key = (id(m), see_emitter)        # m stays alive in the node graph, so its id is stable
            if key in material_cache:
                return material_cache[key]
            p = m.params
            emit = spectrum(m.emitter.emittance) if (m.emitter is not None and see_emitter) else -1
            sp = lambda name: spectrum(p[name].spectrum)
            if m.kind == "inverse":
                raise UnsupportedFeature('an "inverse" material on its own (only as a component of "sum" / "mix")')
            if m.kind in ("sum", "mix"):
                comps, inv = [], []
                for c in (p["mat0"], p["mat1"]):
                    flipped = c.kind == "inverse"
                    leaf = c.params["base"] if flipped else c
                    if leaf.kind == "inverse" or (leaf.kind in ("sum", "mix") and (flipped or multi_levels <= 1)):
                        raise UnsupportedFeature('"sum" / "mix" materials nested more than one level deep, or an "inverse" of one')
                    if flipped and leaf.kind in ("glass", "microfacet glass"):
                        raise UnsupportedFeature('"inverse" of a two-sided material (%s)' % leaf.kind)
                    comps.append(material(leaf, see_emitter=False, multi_levels=multi_levels - 1))      # a component's own emitter is not seen through the sum
                    inv.append(flipped)                                  # (SurfaceMaterial::isEmitting is false for it)
                if m.kind == "sum":
                    idx = b.summed(comps[0], comps[1], inv, emit)
                else:
                    idx = b.mixed(comps[0], comps[1], p["factor"].value, inv, emit)
                material_cache[key] = idx
                return idx
            if m.kind == "matte":
                idx = b.material(abi.MAT_MATTE, (sp("reflectance"), -1, -1), -1.0 if p["sigma"] is None else p["sigma"].value, emit)
            elif m.kind == "metal":
                idx = b.material(abi.MAT_METAL, (sp("coeffR"), sp("eta"), sp("k")), emittance=emit)
            elif m.kind == "glass":
                idx = b.material(abi.MAT_GLASS, (sp("coeff"), sp("etaExt"), sp("etaInt")), emittance=emit)
            elif m.kind == "Ward":
                idx = b.material(abi.MAT_WARD, (sp("R"), -1, -1), p["anisoX"].value, emit, p["anisoY"].value)
            elif m.kind == "Ashikhmin":
                idx = b.material(abi.MAT_ASHIKHMIN, (sp("Rs"), sp("Rd"), -1), p["nx"].value, emit, p["ny"].value)
            elif m.kind == "microfacet metal":
                idx = b.material(abi.MAT_MF_METAL, (-1, sp("eta"), sp("k")), p["alpha_g"].value, emit)
            else:
                idx = b.material(abi.MAT_MF_GLASS, (-1, sp("etaExt"), sp("etaInt")), p["alpha_g"].value, emit)
            material_cache[key] = idx
            return idx

        cameras = []
        mesh_ranges = {}        # referenced node -> its triangles (first, count), flattened once in the node's own space

        def visit(node, world, referenced=False):
            world = world @ node.transform
            for c in node.children:
                if isinstance(c, Node):
                    visit(c, world, referenced)
                elif isinstance(c, ReferenceNode):
                    if referenced:
                        raise UnsupportedFeature("a reference node inside a referenced subtree (one level of instancing)")
                    if id(c.node) not in mesh_ranges:
                        first = b.num_triangles()
                        visit(c.node, np.eye(4), True)
                        count = b.num_triangles() - first
                        if count == 0:
                            raise SceneLanguageError("createReferenceNode of a node without geometry")
                        mesh_ranges[id(c.node)] = (first, count)
                    first, count = mesh_ranges[id(c.node)]
                    b.add_instance(first, count, world)
                elif isinstance(c, Mesh):
                    pos = [v[0] for v in c.vertices]; nrm = [v[1] for v in c.vertices]
                    tan = [v[2] for v in c.vertices]; uv = [v[3] for v in c.vertices]
                    for mat, tris in c.groups:
                        b.add_mesh(pos, nrm, tan, uv, tris, material(mat), world)
                elif isinstance(c, Model):
                    if c.kind == "sphere":
                        b.add_uv_sphere(self.sphere_tessellation[0], self.sphere_tessellation[1], material(c.material), world)
                    else:
                        b.add_box(material(c.material), world @ scenes._scale(2.0))     # models/box.assbin spans [-1, 1]^3
                elif isinstance(c, Camera):
                    if referenced:
                        raise UnsupportedFeature("a camera inside a referenced subtree")
                    cameras.append((c, world))
                else:
                    raise SceneLanguageError("unexpected child in the node graph")

        visit(self.root, np.eye(4))
        if len(cameras) != 1:
            raise SceneLanguageError("the scene needs exactly one camera, found %d" % len(cameras))
        cam, world = cameras[0]
        camera = scenes.make_camera(world, cam.aspect, cam.fovY, cam.radius, cam.imgDist, cam.objDist, cam.sensitivity)
        return b.build(camera, env=environment, name="scene_language")


def _bind_spectrum(b, sv):
    """Spectrum::create (API.cpp:1139-1147 spectral, :1281-1369 RGB) through SceneBuilder: both the RGB-mode value and the
    spectral descriptor."""
    ctor, args, scale = sv.ctor, sv.args, float(sv.scale)
    if ctor == "grey":
        typ, v = args
        if typ != "Reflectance" or scale != 1.0:
            raise UnsupportedFeature("grey spectrum of type %r or scaled" % typ)
        return b.spectrum_grey(v)
    if ctor == "tristimulus":
        typ, space, e0, e1, e2 = args
        if typ != "Reflectance" or space not in ("sRGB", "sRGB_NonLinear") or scale != 1.0:
            raise UnsupportedFeature("tristimulus spectrum %r in space %r" % (typ, space))
        return b.spectrum_srgb_nonlinear(e0, e1, e2)       # "sRGB" means the non-linear space here (API.cpp:62-63,295-296)
    if ctor == "library":
        name, idx = args
        if name == "D65":
            return b.spectrum_d65(scale, scenes.D65_RGB)
        from . import spectra
        if ("ior_%s_meta" % name) not in spectra.tables() or idx not in (0, 1) or scale != 1.0:
            raise UnsupportedFeature("library spectrum %r[%d] (scaled or not in spectrum_library.cpp)" % (name, idx))
        return b.spectrum_ior(name, idx, spectra.named_rgb(name, idx))      # its RGB-build value: API.cpp:1216-1278,1326-1369
    raise UnsupportedFeature("spectrum constructor %r" % ctor)


def load_scene(path_or_text, environment=None, sphere_tessellation=(48, 24)):
    """Runs a scene script and returns (abi.Scene, render settings dict, renderer dict)."""
    text = path_or_text
    if "\n" not in path_or_text and not path_or_text.rstrip().endswith(";"):
        with open(path_or_text) as fh:
            text = fh.read()
    it = Interpreter(sphere_tessellation)
    it.run(text)
    return it.build(environment), dict(it.settings), dict(it.renderer)
