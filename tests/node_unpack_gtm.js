// Test driver (ours): loads the reference's lzma.js + lzma.shim.js from the directory given as argv[2] and unpacks the
// keyframe streams of a .gtm file the way its web worker does (one LZMA.decompressFile per stream until the input
// ends).  Writes the concatenated raw command bytes to argv[4] and prints each stream's raw size.
const fs = require('fs');
const vm = require('vm');
const path = require('path');
const [dir, inPath, outPath] = process.argv.slice(2);
const ctx = vm.createContext({ Uint8Array, ArrayBuffer, console });
for (const f of ['lzma.js', 'lzma.shim.js']) vm.runInContext(fs.readFileSync(path.join(dir, f), 'utf8'), ctx, { filename: f });
const LZMA = vm.runInContext('LZMA', ctx);
const file = fs.readFileSync(inPath);
const whole = file.readUInt32LE(8);
const body = file.subarray(whole);
const ab = new ArrayBuffer(body.length);
new Uint8Array(ab).set(body);
const inStream = new LZMA.iStream(ab);
const chunks = [];
const sizes = [];
while (inStream.offset < inStream.size) {
  const outStream = new LZMA.oStream();
  LZMA.decompressFile(inStream, outStream);
  const u8 = outStream.toUint8Array();
  chunks.push(Buffer.from(u8));
  sizes.push(u8.length);
}
fs.writeFileSync(outPath, Buffer.concat(chunks));
console.log(sizes.join(' '));
