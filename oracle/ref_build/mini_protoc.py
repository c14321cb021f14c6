#!/usr/bin/env python3
"""mini_protoc.py -- .proto (proto3) -> header-only C++ structs with protoc's ACCESSOR API.

TEST INFRASTRUCTURE (oracle/): part of the recipe that compiles the REFERENCE's own encoder sources
from where they lie under /root/reference into oracle/_ref/libdvref.so (see oracle/ref_build/README.md).
The reference's C++ is written against protoc-generated classes (`deepvariant/protos/deepvariant.pb.h`,
`third_party/nucleus/protos/*.pb.h`); neither protoc nor libprotobuf is in this image.  Its sources use
only the generated ACCESSORS (`read.alignment().position().position()`, `call.allele_support().at(a)
.read_names()`, `value.kind_case()`, enum constants), never serialisation or reflection -- so a plain
struct per message with the same accessor names and semantics is enough to compile them unmodified.

For every message: scalar `T f() const / set_f / clear_f`, string `const std::string& f() / set_f /
mutable_f`, message `const M& f() / mutable_f / has_f / clear_f`, repeated `f() / f(i) / f_size / add_f /
mutable_f`, map `f() / mutable_f / f_size`, oneof `<name>_case()` with `kCamelCase` constants; nested types
as `Outer_Inner` with `Outer::Inner` aliases; enums with `Enum_Name()`.  Proto-default semantics (zero /
empty / default instance for unset message fields; reading a oneof member that is not set gives the
default).  Runtime pieces (RepeatedField, RepeatedPtrField, Map, Box) live in
oracle/ref_build/shims/google/protobuf/mini_runtime.h.

Usage: mini_protoc.py --root /root/reference --out oracle/_ref/gen  a.proto b.proto ...   (imports follow)
"""
import argparse
import os
import re
import sys

SCALARS = {
    'double': 'double', 'float': 'float', 'int32': 'int32_t', 'int64': 'int64_t', 'uint32': 'uint32_t',
    'uint64': 'uint64_t', 'sint32': 'int32_t', 'sint64': 'int64_t', 'fixed32': 'uint32_t', 'fixed64': 'uint64_t',
    'sfixed32': 'int32_t', 'sfixed64': 'int64_t', 'bool': 'bool',
}
STRINGS = ('string', 'bytes')


def tokenize(text):
  text = re.sub(r'/\*.*?\*/', ' ', text, flags=re.S)
  text = re.sub(r'//[^\n]*', ' ', text)
  return re.findall(r'"(?:[^"\\]|\\.)*"|\'(?:[^\'\\]|\\.)*\'|[A-Za-z_][A-Za-z0-9_.]*|-?[0-9][0-9a-fA-FxX.]*|[{}\[\]()<>=;,]', text)


class Enum:
  def __init__(self, name, scope):
    self.name, self.scope, self.values = name, scope, []      # scope: list of enclosing message names


class Field:
  def __init__(self):
    self.label = self.type = self.name = None
    self.number = 0
    self.map_key = self.map_value = None
    self.oneof = None
    self.optional = False


class Message:
  def __init__(self, name, scope):
    self.name, self.scope = name, scope
    self.fields, self.messages, self.enums, self.oneofs = [], [], [], []


class File:
  def __init__(self, path):
    self.path, self.package, self.imports, self.messages, self.enums = path, '', [], [], []


class Parser:
  def __init__(self, toks):
    self.t, self.i = toks, 0

  def peek(self):
    return self.t[self.i] if self.i < len(self.t) else None

  def next(self):
    tok = self.t[self.i]
    self.i += 1
    return tok

  def expect(self, tok):
    got = self.next()
    if got != tok:
      raise SyntaxError('expected %r, got %r near token %d' % (tok, got, self.i))

  def skip_statement(self):
    depth = 0
    while True:
      tok = self.next()
      if tok in '{[(':
        depth += 1
      elif tok in '}])':
        depth -= 1
      elif tok == ';' and depth == 0:
        return

  def skip_options(self):
    if self.peek() == '[':
      depth = 0
      while True:
        tok = self.next()
        if tok == '[':
          depth += 1
        elif tok == ']':
          depth -= 1
          if depth == 0:
            return

  def file(self, path):
    f = File(path)
    while self.peek() is not None:
      tok = self.peek()
      if tok == 'syntax':
        self.skip_statement()
      elif tok == 'package':
        self.next()
        f.package = self.next()
        self.expect(';')
      elif tok == 'import':
        self.next()
        if self.peek() in ('public', 'weak'):
          self.next()
        f.imports.append(self.next().strip('"'))
        self.expect(';')
      elif tok == 'option':
        self.skip_statement()
      elif tok == 'message':
        f.messages.append(self.message([]))
      elif tok == 'enum':
        f.enums.append(self.enum([]))
      elif tok == ';':
        self.next()
      else:
        raise SyntaxError('unexpected %r at top level of %s' % (tok, path))
    return f

  def enum(self, scope):
    self.expect('enum')
    e = Enum(self.next(), scope)
    self.expect('{')
    while self.peek() != '}':
      tok = self.peek()
      if tok in ('option', 'reserved'):
        self.skip_statement()
        continue
      if tok == ';':
        self.next()
        continue
      name = self.next()
      self.expect('=')
      e.values.append((name, int(self.next(), 0)))
      self.skip_options()
      self.expect(';')
    self.expect('}')
    return e

  def field(self, m, oneof=None):
    f = Field()
    f.oneof = oneof
    tok = self.next()
    if tok in ('repeated', 'optional', 'required'):
      f.label = tok if tok == 'repeated' else None
      f.optional = tok == 'optional'
      tok = self.next()
    if tok == 'map':
      self.expect('<')
      f.map_key = self.next()
      self.expect(',')
      f.map_value = self.next()
      self.expect('>')
      f.type = 'map'
    else:
      f.type = tok
    f.name = self.next()
    self.expect('=')
    f.number = int(self.next(), 0)
    self.skip_options()
    self.expect(';')
    m.fields.append(f)

  def message(self, scope):
    self.expect('message')
    m = Message(self.next(), scope)
    inner = scope + [m.name]
    self.expect('{')
    while self.peek() != '}':
      tok = self.peek()
      if tok == 'message':
        m.messages.append(self.message(inner))
      elif tok == 'enum':
        m.enums.append(self.enum(inner))
      elif tok in ('option', 'reserved', 'extensions'):
        self.skip_statement()
      elif tok == 'oneof':
        self.next()
        name = self.next()
        m.oneofs.append(name)
        self.expect('{')
        while self.peek() != '}':
          if self.peek() == 'option':
            self.skip_statement()
          else:
            self.field(m, oneof=name)
        self.expect('}')
      elif tok == ';':
        self.next()
      else:
        self.field(m)
    self.expect('}')
    return m



def wire_scalar(t):
  """proto scalar type -> (wire type, C++ expression template writing value `v` into *out)."""
  W = '::google::protobuf::mini::'
  if t in ('int32', 'int64', 'uint32', 'uint64', 'bool'):
    return 0, W + 'PutVarint(out, static_cast<uint64_t>(static_cast<int64_t>(%s)))' if t in ('int32', 'int64') else W + 'PutVarint(out, static_cast<uint64_t>(%s))'
  if t in ('sint32', 'sint64'):
    return 0, W + 'PutVarint(out, ' + W + 'ZigZag(static_cast<int64_t>(%s)))'
  if t in ('fixed32', 'sfixed32', 'float'):
    return 5, W + 'PutFixed(out, %s)'
  if t in ('fixed64', 'sfixed64', 'double'):
    return 1, W + 'PutFixed(out, %s)'
  raise KeyError(t)


def camel(name):
  return ''.join(p[:1].upper() + p[1:] for p in name.split('_'))


class Generator:
  def __init__(self, root):
    self.roots = root if isinstance(root, (list, tuple)) else [root]
    self.files = {}           # path -> File
    self.types = {}           # fully qualified proto name -> ('message' | 'enum', cpp namespace, cpp class name)

  def load(self, path):
    if path in self.files:
      return
    for root in self.roots:
      if os.path.exists(os.path.join(root, path)):
        break
    else:
      raise FileNotFoundError(path)
    with open(os.path.join(root, path)) as fh:
      f = Parser(tokenize(fh.read())).file(path)
    self.files[path] = f
    ns = '::'.join(f.package.split('.')) if f.package else ''

    def register(m):
      fq = '.'.join([f.package] + m.scope + [m.name]) if f.package else '.'.join(m.scope + [m.name])
      self.types[fq] = ('message', ns, '_'.join(m.scope + [m.name]))
      for e in m.enums:
        self.types[fq + '.' + e.name] = ('enum', ns, '_'.join(e.scope + [e.name]))
      for sub in m.messages:
        register(sub)
    for m in f.messages:
      register(m)
    for e in f.enums:
      self.types[(f.package + '.' if f.package else '') + e.name] = ('enum', ns, e.name)
    for imp in f.imports:
      self.load(imp)

  def resolve(self, f, scope, name):
    """proto name as written inside `scope` of file f -> (kind, fully qualified C++ name)."""
    if name.startswith('.'):
      cands = [name[1:]]
    else:
      pkg = f.package.split('.') if f.package else []
      chain = pkg + scope
      cands = ['.'.join(chain[:k] + [name]) for k in range(len(chain), -1, -1)]
    for c in cands:
      if c in self.types:
        kind, ns, cls = self.types[c]
        return kind, ('::' + ns + '::' + cls) if ns else ('::' + cls)
    raise KeyError('cannot resolve type %s in %s (%s)' % (name, f.path, '.'.join(scope)))

  # ---- C++ emission
  def cpp_type(self, f, scope, t):
    if t in SCALARS:
      return 'scalar', SCALARS[t]
    if t in STRINGS:
      return 'string', 'std::string'
    kind, cpp = self.resolve(f, scope, t)
    return kind, cpp

  def emit(self, path, out_dir):
    f = self.files[path]
    guard = re.sub(r'[^A-Za-z0-9]', '_', path).upper() + '_MINI_PB_H_'
    o = []
    w = o.append
    w('// Generated by oracle/ref_build/mini_protoc.py from %s -- plain structs with protoc\'s accessor API.' % path)
    w('// TEST INFRASTRUCTURE: lets the reference\'s own sources compile without protoc / libprotobuf.')
    w('#ifndef %s\n#define %s' % (guard, guard))
    w('#include <cstdint>\n#include <string>\n#include "google/protobuf/mini_runtime.h"')
    for imp in f.imports:
      w('#include "%s"' % imp.replace('.proto', '.pb.h'))
    ns_parts = f.package.split('.') if f.package else []
    for p in ns_parts:
      w('namespace %s {' % p)

    all_msgs = []

    def walk(m):
      all_msgs.append(m)
      for sub in m.messages:
        walk(sub)
    for m in f.messages:
      walk(m)
    all_enums = list(f.enums) + [e for m in all_msgs for e in m.enums]
    # enums
    for e in all_enums:
      cls = '_'.join(e.scope + [e.name])
      prefix = cls + '_' if e.scope else ''
      w('enum %s : int {' % cls)
      for name, num in e.values:
        w('  %s%s = %d,' % (prefix, name, num))
      w('};')
      w('inline const std::string& %s_Name(int v) {' % cls)
      w('  static const std::string kUnknown;')
      seen = set()
      for name, num in e.values:
        if num in seen:
          continue
        seen.add(num)
        w('  if (v == %d) { static const std::string s = "%s"; return s; }' % (num, name))
      w('  return kUnknown;\n}')
      w('inline bool %s_IsValid(int v) { return %s; }' % (cls, ' || '.join('v == %d' % n for n in sorted({n for _, n in e.values})) or 'false'))
      w('inline const ::google::protobuf::mini::EnumDescriptor* %s_descriptor() {' % cls)
      w('  static const ::google::protobuf::mini::EnumDescriptor* d = new ::google::protobuf::mini::EnumDescriptor({%s});'
        % ', '.join('{"%s", %d}' % (name, num) for name, num in e.values))
      w('  return d;\n}')
    # forward declarations
    for m in all_msgs:
      w('class %s;' % '_'.join(m.scope + [m.name]))
    # definition order: a message after the messages it holds BY VALUE (repeated elements, map values)
    by_name = {'_'.join(m.scope + [m.name]): m for m in all_msgs}
    local = lambda cpp: cpp.split('::')[-1] if cpp.rsplit('::', 1)[0].strip(':') == '::'.join(ns_parts) else None   # noqa: E731

    def value_deps(m):
      deps = []
      for fld in m.fields:
        t = fld.map_value if fld.type == 'map' else (fld.type if fld.label == 'repeated' else None)
        if t and t not in SCALARS and t not in STRINGS:
          kind, cpp = self.resolve(f, m.scope + [m.name], t)
          if kind == 'message' and local(cpp) in by_name:
            deps.append(local(cpp))
      return deps
    ordered, state = [], {}

    def visit(name):
      if state.get(name) == 2:
        return
      if state.get(name) == 1:
        raise ValueError('by-value cycle through %s in %s' % (name, path))
      state[name] = 1
      for d in value_deps(by_name[name]):
        visit(d)
      state[name] = 2
      ordered.append(by_name[name])
    for name in by_name:
      visit(name)

    bodies = []
    W = '::google::protobuf::mini::'

    def put_value(scope, t, expr):
      """statement(s) writing one value of proto type t (no tag) -- scalars, strings, enums, messages."""
      if t in SCALARS:
        return wire_scalar(t)[1] % expr + ';'
      if t in STRINGS:
        return W + 'PutBytes(out, %s);' % expr
      kind, _ = self.resolve(f, scope, t)
      if kind == 'enum':
        return W + 'PutVarint(out, static_cast<uint64_t>(static_cast<int64_t>(%s)));' % expr
      return '{ std::string sub; (%s).AppendTo(&sub); %sPutBytes(out, sub); }' % (expr, W)

    def wire_type(scope, t):
      if t in SCALARS:
        return wire_scalar(t)[0]
      if t in STRINGS:
        return 2
      kind, _ = self.resolve(f, scope, t)
      return 0 if kind == 'enum' else 2

    def tag(number, wt):
      return W + 'PutVarint(out, %du);' % ((number << 3) | wt)
    for m in ordered:
      cls = '_'.join(m.scope + [m.name])
      scope = m.scope + [m.name]
      w('class %s {\n public:' % cls)
      for sub in m.messages:
        w('  using %s = %s_%s;' % (sub.name, cls, sub.name))
      for e in m.enums:
        ecls = cls + '_' + e.name
        w('  using %s = %s;' % (e.name, ecls))
        for name, _ in e.values:
          w('  static constexpr %s %s = %s_%s;' % (e.name, name, ecls, name))
        w('  static const std::string& %s_Name(int v) { return %s_Name(v); }' % (e.name, ecls))
      for oneof in m.oneofs:
        w('  enum %sCase {' % camel(oneof))
        for fld in m.fields:
          if fld.oneof == oneof:
            w('    k%s = %d,' % (camel(fld.name), fld.number))
        w('    %s_NOT_SET = 0,\n  };' % oneof.upper())
        w('  %sCase %s_case() const { return static_cast<%sCase>(_oneof_%s_); }' % (camel(oneof), oneof, camel(oneof), oneof))
        w('  void clear_%s() { _oneof_%s_ = 0; }' % (oneof, oneof))
      members = []
      merges = []      # statements of MergeFrom(const cls& o)
      writes = []      # (field number, statement of AppendTo(std::string* out))
      for fld in m.fields:
        n = fld.name
        guard_get = guard_set = ''
        if fld.oneof:
          guard_get = '_oneof_%s_ == %d' % (fld.oneof, fld.number)
          guard_set = '_oneof_%s_ = %d; ' % (fld.oneof, fld.number)
          w('  bool has_%s() const { return %s; }' % (n, guard_get))
        if fld.type == 'map':
          _, kt = self.cpp_type(f, scope, fld.map_key)
          _, vt = self.cpp_type(f, scope, fld.map_value)
          mt = '::google::protobuf::Map<%s, %s>' % (kt, vt)
          w('  const %s& %s() const { return %s_; }' % (mt, n, n))
          w('  %s* mutable_%s() { return &%s_; }' % (mt, n, n))
          w('  int %s_size() const { return static_cast<int>(%s_.size()); }' % (n, n))
          w('  void clear_%s() { %s_.clear(); }' % (n, n))
          members.append('  %s %s_;' % (mt, n))
          merges.append('for (const auto& kv : o.%s_) %s_[kv.first] = kv.second;' % (n, n))
          writes.append((fld.number, 'for (const auto& kv : %s_) { std::string e; { std::string* out = &e; %s %s %s %s } %s %sPutBytes(out, e); }' % (
              n, tag(1, wire_type(scope, fld.map_key)), put_value(scope, fld.map_key, 'kv.first'),
              tag(2, wire_type(scope, fld.map_value)), put_value(scope, fld.map_value, 'kv.second'), tag(fld.number, 2), W)))
          continue
        kind, ct = self.cpp_type(f, scope, fld.type)
        if fld.label == 'repeated':
          if kind in ('scalar', 'enum'):
            rt = '::google::protobuf::RepeatedField<%s>' % ct
            w('  %s %s(int i) const { return %s_[i]; }' % (ct, n, n))
            w('  void add_%s(%s v) { %s_.push_back(v); }' % (n, ct, n))
            w('  void set_%s(int i, %s v) { %s_[i] = v; }' % (n, ct, n))
          elif kind == 'string':
            rt = '::google::protobuf::RepeatedPtrField<std::string>'
            w('  const std::string& %s(int i) const { return %s_[i]; }' % (n, n))
            w('  void add_%s(const std::string& v) { %s_.push_back(v); }' % (n, n))
            w('  void add_%s(const char* v) { %s_.emplace_back(v); }' % (n, n))
            w('  void add_%s(const void* v, size_t len) { %s_.emplace_back(static_cast<const char*>(v), len); }' % (n, n))
            w('  void add_%s(std::string&& v) { %s_.push_back(std::move(v)); }' % (n, n))
            w('  std::string* add_%s() { %s_.emplace_back(); return &%s_.back(); }' % (n, n, n))
            w('  std::string* mutable_%s(int i) { return &%s_[i]; }' % (n, n))
            w('  void set_%s(int i, const std::string& v) { %s_[i] = v; }' % (n, n))
          else:
            rt = '::google::protobuf::RepeatedPtrField<%s>' % ct
            w('  const %s& %s(int i) const;' % (ct, n))
            w('  %s* add_%s();' % (ct, n))
            w('  %s* mutable_%s(int i);' % (ct, n))
            bodies.append('inline const %s& %s::%s(int i) const { return %s_[i]; }' % (ct, cls, n, n))
            bodies.append('inline %s* %s::add_%s() { %s_.emplace_back(); return &%s_.back(); }' % (ct, cls, n, n, n))
            bodies.append('inline %s* %s::mutable_%s(int i) { return &%s_[i]; }' % (ct, cls, n, n))
          w('  const %s& %s() const { return %s_; }' % (rt, n, n))
          w('  %s* mutable_%s() { return &%s_; }' % (rt, n, n))
          w('  int %s_size() const { return static_cast<int>(%s_.size()); }' % (n, n))
          w('  void clear_%s() { %s_.clear(); }' % (n, n))
          members.append('  %s %s_;' % (rt, n))
          merges.append('for (const auto& e : o.%s_) %s_.push_back(e);' % (n, n))
          if kind in ('scalar', 'enum'):      # packed (proto3)
            writes.append((fld.number, 'if (!%s_.empty()) { std::string p; { std::string* out = &p; for (const auto& e : %s_) { %s } } %s %sPutBytes(out, p); }' % (
                n, n, put_value(scope, fld.type, 'e'), tag(fld.number, 2), W)))
          else:
            writes.append((fld.number, 'for (const auto& e : %s_) { %s %s }' % (n, tag(fld.number, 2), put_value(scope, fld.type, 'e'))))
        elif kind in ('scalar', 'enum'):
          zero = 'static_cast<%s>(0)' % ct
          if fld.oneof:
            w('  %s %s() const { return %s ? %s_ : %s; }' % (ct, n, guard_get, n, zero))
          else:
            w('  %s %s() const { return %s_; }' % (ct, n, n))
          tail = ' _has_%s_ = true;' % n if fld.optional else ''
          w('  void set_%s(%s v) { %s%s_ = v;%s }' % (n, ct, guard_set, n, tail))
          w('  void clear_%s() { %s_ = %s;%s }' % (n, n, zero, (' _has_%s_ = false;' % n) if fld.optional else ''))
          if fld.optional:
            w('  bool has_%s() const { return _has_%s_; }' % (n, n))
            members.append('  bool _has_%s_ = false;' % n)
          members.append('  %s %s_ = %s;' % (ct, n, zero))
          if fld.oneof:
            merges.append('if (o.%s) set_%s(o.%s_);' % (guard_get, n, n))
            cond = guard_get
          elif fld.optional:
            merges.append('if (o._has_%s_) set_%s(o.%s_);' % (n, n, n))
            cond = '_has_%s_' % n
          else:
            merges.append('if (o.%s_ != %s) %s_ = o.%s_;' % (n, zero, n, n))
            cond = '%s_ != %s' % (n, zero)
          writes.append((fld.number, 'if (%s) { %s %s }' % (cond, tag(fld.number, wire_type(scope, fld.type)), put_value(scope, fld.type, n + '_'))))
        elif kind == 'string':
          if fld.oneof:
            w('  const std::string& %s() const { return %s ? %s_ : ::google::protobuf::mini::EmptyString(); }' % (n, guard_get, n))
          else:
            w('  const std::string& %s() const { return %s_; }' % (n, n))
          tail = ' _has_%s_ = true;' % n if fld.optional else ''
          w('  void set_%s(const std::string& v) { %s%s_ = v;%s }' % (n, guard_set, n, tail))
          w('  void set_%s(std::string&& v) { %s%s_ = std::move(v);%s }' % (n, guard_set, n, tail))
          w('  void set_%s(const char* v) { %s%s_ = v;%s }' % (n, guard_set, n, tail))
          w('  void set_%s(const char* v, size_t len) { %s%s_.assign(v, len);%s }' % (n, guard_set, n, tail))
          w('  void set_%s(std::string_view v) { %s%s_.assign(v.data(), v.size());%s }' % (n, guard_set, n, tail))
          w('  std::string* mutable_%s() { %s%sreturn &%s_; }' % (n, guard_set, tail.strip() + ' ' if tail else '', n))
          w('  void clear_%s() { %s_.clear(); }' % (n, n))
          if fld.optional:
            w('  bool has_%s() const { return _has_%s_; }' % (n, n))
            members.append('  bool _has_%s_ = false;' % n)
          members.append('  std::string %s_;' % n)
          if fld.oneof:
            merges.append('if (o.%s) set_%s(o.%s_);' % (guard_get, n, n))
            cond = guard_get
          elif fld.optional:
            merges.append('if (o._has_%s_) set_%s(o.%s_);' % (n, n, n))
            cond = '_has_%s_' % n
          else:
            merges.append('if (!o.%s_.empty()) %s_ = o.%s_;' % (n, n, n))
            cond = '!%s_.empty()' % n
          writes.append((fld.number, 'if (%s) { %s %sPutBytes(out, %s_); }' % (cond, tag(fld.number, 2), W, n)))
        else:     # singular message: held through a copying pointer (the type may still be incomplete here)
          w('  const %s& %s() const;' % (ct, n))
          w('  %s* mutable_%s();' % (ct, n))
          w('  void clear_%s();' % n)
          if fld.oneof:
            bodies.append('inline const %s& %s::%s() const { return %s ? %s_.get() : ::google::protobuf::mini::Default<%s>(); }'
                          % (ct, cls, n, guard_get, n, ct))
          else:
            w('  bool has_%s() const { return %s_.has(); }' % (n, n))
            bodies.append('inline const %s& %s::%s() const { return %s_.get(); }' % (ct, cls, n, n))
          bodies.append('inline %s* %s::mutable_%s() { %sreturn %s_.mutable_get(); }' % (ct, cls, n, guard_set, n))
          bodies.append('inline void %s::clear_%s() { %s_.reset(); }' % (cls, n, n))
          w('  void set_allocated_%s(%s* p);' % (n, ct))
          w('  %s* release_%s();' % (ct, n))
          bodies.append('inline void %s::set_allocated_%s(%s* p) { %s%s_.adopt(p); }' % (cls, n, ct, guard_set, n))
          bodies.append('inline %s* %s::release_%s() { return %s_.release(); }' % (ct, cls, n, n))
          members.append('  ::google::protobuf::mini::Box<%s> %s_;' % (ct, n))
          if fld.oneof:
            merges.append('if (o.%s) mutable_%s()->MergeFrom(o.%s_.get());' % (guard_get, n, n))
            cond = guard_get
          else:
            merges.append('if (o.%s_.has()) mutable_%s()->MergeFrom(o.%s_.get());' % (n, n, n))
            cond = '%s_.has()' % n
          writes.append((fld.number, 'if (%s) { %s %s }' % (cond, tag(fld.number, 2), put_value(scope, fld.type, n + '_.get()'))))
      # (text format is only ever used inside the reference's log / CHECK messages)
      w('  std::string DebugString() const { return "<%s>"; }' % cls)
      w('  std::string ShortDebugString() const { return "<%s>"; }' % cls)
      w('  void Clear() { *this = %s(); }' % cls)
      w('  void CopyFrom(const %s& other) { *this = other; }' % cls)
      w('  void MergeFrom(const %s& o);' % cls)
      # wire format (fields in number order, packed repeated scalars, map entries in key order): what the
      # reference's EncodeExample hands to its TFRecord writer
      w('  void AppendTo(std::string* out) const;')
      w('  bool SerializeToString(std::string* out) const { out->clear(); AppendTo(out); return true; }')
      w('  std::string SerializeAsString() const { std::string s; AppendTo(&s); return s; }')
      w('  size_t ByteSizeLong() const { return SerializeAsString().size(); }')
      bodies.append('inline void %s::AppendTo(std::string* out) const {\n  %s\n}' % (
          cls, '\n  '.join(stmt for _, stmt in sorted(writes, key=lambda t: t[0])) or '(void)out;'))
      bodies.append('inline void %s::MergeFrom(const %s& o) {\n  %s\n}' % (cls, cls, '\n  '.join(merges) or '(void)o;'))
      w(' private:')
      for oneof in m.oneofs:
        members.append('  int _oneof_%s_ = 0;' % oneof)
      o.extend(members)
      w('};')
    o.extend(bodies)
    for p in reversed(ns_parts):
      w('}  // namespace %s' % p)
    w('#endif  // %s' % guard)
    dst = os.path.join(out_dir, path.replace('.proto', '.pb.h'))
    os.makedirs(os.path.dirname(dst), exist_ok=True)
    with open(dst, 'w') as fh:
      fh.write('\n'.join(o) + '\n')
    return dst


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--root', required=True, action='append')
  ap.add_argument('--out', required=True)
  ap.add_argument('protos', nargs='+')
  a = ap.parse_args()
  g = Generator(a.root)
  for p in a.protos:
    g.load(p)
  for path in g.files:
    g.emit(path, a.out)
  print('mini_protoc: %d files -> %s' % (len(g.files), a.out), file=sys.stderr)


if __name__ == '__main__':
  main()
