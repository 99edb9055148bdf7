// oracle/ply_ref.mjs — TEST INFRASTRUCTURE.  Runs the REFERENCE's own PLY header / vertex decoding
// (/root/reference/src/loaders/ply/PlyParserUtils.js is three-free and imports under Node) on a file and dumps what it
// derives.  The field-name -> id map is the one INRIAV1PlyParser.decodeHeaderLines builds (that module imports 'three',
// so its 20 lines of map construction are restated here from INRIAV1PlyParser.js:7-47).
// usage: node ply_ref.mjs <PlyParserUtils.js> <file.ply> <out.json> <rows>
import fs from 'fs';
const [utilsPath, plyPath, outPath, rowsArg] = process.argv.slice(2);
import(utilsPath).then(({ PlyParserUtils }) => {
const buf = fs.readFileSync(plyPath);
const ab = buf.buffer.slice(buf.byteOffset, buf.byteOffset + buf.byteLength);
const headerText = PlyParserUtils.readHeaderFromBuffer(ab);
const headerLines = PlyParserUtils.convertHeaderTextToLines(headerText);
const base = ['scale_0', 'scale_1', 'scale_2', 'rot_0', 'rot_1', 'rot_2', 'rot_3', 'x', 'y', 'z',
              'f_dc_0', 'f_dc_1', 'f_dc_2', 'opacity', 'red', 'green', 'blue', 'f_rest_0'];
let shLineCount = 0;
headerLines.forEach((line) => { if (line.includes('f_rest_')) shLineCount++; });
let shFieldsToReadCount = 0;
if (shLineCount >= 45) shFieldsToReadCount = 45; else if (shLineCount >= 24) shFieldsToReadCount = 24; else if (shLineCount >= 9) shFieldsToReadCount = 9;
const rest = Array.from(Array(Math.max(shFieldsToReadCount - 1, 0))).map((e, i) => `f_rest_${i + 1}`);
const names = [...base, ...rest];
const map = {}; names.forEach((n, i) => { map[n] = i; });
const header = PlyParserUtils.decodeSectionHeader(headerLines, map, 0);
const headerSizeBytes = headerText.indexOf(PlyParserUtils.HeaderEndToken) + PlyParserUtils.HeaderEndToken.length + 1;
const view = new DataView(ab, headerSizeBytes);
const rows = [];
const ids = names.map((n, i) => i);
for (let r = 0; r < Math.min(parseInt(rowsArg), header.vertexCount); r++) {
  const raw = [];
  PlyParserUtils.readVertex(view, header, r, 0, ids, raw, true);
  const o = {};
  names.forEach((n, i) => { if (raw[i] !== undefined) o[n] = raw[i]; });
  rows.push(o);
}
fs.writeFileSync(outPath, JSON.stringify({
  vertexCount: header.vertexCount, bytesPerVertex: header.bytesPerVertex, headerSizeBytes: headerSizeBytes,
  sphericalHarmonicsDegree: header.sphericalHarmonicsDegree, coefficientsPerChannel: header.sphericalHarmonicsCoefficientsPerChannel,
  degree1Fields: header.sphericalHarmonicsDegree1Fields.map((id) => names[id]),
  degree2Fields: header.sphericalHarmonicsDegree2Fields.map((id) => names[id]),
  fieldOffsets: Object.fromEntries(names.map((n, i) => [n, header.fieldOffsets[i]]).filter((e) => e[1] !== undefined)),
  rows: rows }));
console.log('ok');
}).catch((e) => { console.error(String(e && e.stack || e)); process.exit(1); });
