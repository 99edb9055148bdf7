// oracle/three_min_check.mjs — TEST INFRASTRUCTURE: evaluates the primitives of three_min.mjs on the inputs in <in.json> so
// that tests/test_three_min.py can compare them with independent numpy arithmetic.  usage: node three_min_check.mjs in.json
import fs from 'fs';
import * as THREE from './three_min.mjs';
const inp = JSON.parse(fs.readFileSync(process.argv[2], 'utf8'));
const M4 = (a) => new THREE.Matrix4().fromArray(a);
const out = { invert: [], multiply: [], premultiply: [], applyMatrix4: [], compose: [], decompose: [], m3: [], normalize: [], quatNormalize: [],
              determinant: [], toHalf: [], fromHalf: [] };
for (const c of inp.mats) {
  out.invert.push(M4(c.a).invert().elements);
  out.multiply.push(M4(c.a).multiply(M4(c.b)).elements);
  out.premultiply.push(M4(c.a).premultiply(M4(c.b)).elements);
  out.determinant.push(M4(c.a).determinant());
  out.applyMatrix4.push(new THREE.Vector3(...c.v).applyMatrix4(M4(c.a)).toArray());
  const q = new THREE.Quaternion(...c.q).normalize();
  out.quatNormalize.push([q.x, q.y, q.z, q.w]);
  const m = new THREE.Matrix4().compose(new THREE.Vector3(...c.v), q, new THREE.Vector3(...c.s));
  out.compose.push(m.elements.slice());
  const p = new THREE.Vector3(), r = new THREE.Quaternion(), s = new THREE.Vector3();
  m.decompose(p, r, s);
  out.decompose.push([p.x, p.y, p.z, r.x, r.y, r.z, r.w, s.x, s.y, s.z]);
  const a3 = new THREE.Matrix3().setFromMatrix4(M4(c.a)), b3 = new THREE.Matrix3().setFromMatrix4(M4(c.b));
  out.m3.push({ mul: new THREE.Matrix3().copy(a3).multiply(b3).elements, pre: new THREE.Matrix3().copy(a3).premultiply(b3).elements,
                tr: new THREE.Matrix3().copy(a3).transpose().elements });
  out.normalize.push(new THREE.Vector3(...c.v).normalize().toArray());
}
out.zeroNormalize = new THREE.Vector3(0, 0, 0).normalize().toArray();
out.zeroQuat = (() => { const q = new THREE.Quaternion(0, 0, 0, 0).normalize(); return [q.x, q.y, q.z, q.w]; })();
for (const f of inp.floats) out.toHalf.push(THREE.DataUtils.toHalfFloat(f));
for (let h = 0; h < 65536; h += 1) out.fromHalf.push(THREE.DataUtils.fromHalfFloat(h));
out.perspective = new THREE.Matrix4().makePerspective(-0.2, 0.2, 0.1, -0.1, 0.1, 1000).elements;
console.log(JSON.stringify(out, (k, v) => (typeof v === 'number' && !isFinite(v)) ? String(v) : v));
