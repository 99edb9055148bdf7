// oracle/three_loader.mjs — TEST INFRASTRUCTURE.  ESM resolve hook: the bare specifier 'three' (an uninstalled peer
// dependency of the reference, package.json:69-71) resolves to oracle/three_min.mjs, so the reference's modules import and
// run under Node unmodified, in place under /root/reference.  usage: node --experimental-loader ./three_loader.mjs x.mjs
import { pathToFileURL, fileURLToPath } from 'url';
import path from 'path';
const shim = pathToFileURL(path.join(path.dirname(fileURLToPath(import.meta.url)), 'three_min.mjs')).href;
export async function resolve(specifier, context, defaultResolve) {
  if (specifier === 'three') return { url: shim };
  return defaultResolve(specifier, context, defaultResolve);
}
