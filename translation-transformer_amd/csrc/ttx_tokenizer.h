// Host-side SMILES tokenizer, collate and detokenizer of libttx_hip.so (SURVEY.md §8(f) "next" #2): the string work
// either side of the hot path, which becomes the end-to-end bottleneck once decoding runs at thousands of
// reactions per second.  Restates, as a hand-written left-to-right scanner, the reference's regular expression
//   (\[[^\]]+]|Br?|Cl?|N|O|S|P|F|I|b|c|n|o|s|p|\(|\)|\.|=|#|-|\+|\\|\/|:|~|@|\?|>|\*|\$|\%[0-9]{2}|[0-9])
// (src/data_handling/tokenizer_smiles.py:8) with re.findall semantics (characters no alternative matches are
// skipped), ChemSMILESTokenizer.encode (:34-39: BOS + ids + EOS, unknown pieces -> UNK) and
// GenericTokenizer.decode (tokenizer_base.py:80-91).  No GPU involved.
#pragma once
#include <cstdint>
#include <cstring>
#include <string>
#include <unordered_map>
#include <vector>

struct ttx_tokenizer {
  std::unordered_map<std::string, int32_t> enc;
  std::vector<std::string> dec;            // id -> token ("" when the id is unused)
  int32_t pad = 0, bos = 1, eos = 2, unk = 3;   // tokenizer_base.py:27-30
};

namespace ttxtok {

// Length of the token starting at s[i] (0: no alternative matches there).
inline size_t match(const char* s, size_t n, size_t i) {
  const unsigned char c = (unsigned char)s[i];
  switch (c) {
    case '[': {                             // \[[^\]]+]  — at least one character before the closing bracket
      size_t j = i + 1;
      while (j < n && s[j] != ']') ++j;
      return (j < n && j > i + 1) ? j - i + 1 : 0;
    }
    case 'B': return (i + 1 < n && s[i + 1] == 'r') ? 2 : 1;
    case 'C': return (i + 1 < n && s[i + 1] == 'l') ? 2 : 1;
    case 'N': case 'O': case 'S': case 'P': case 'F': case 'I':
    case 'b': case 'c': case 'n': case 'o': case 's': case 'p':
    case '(': case ')': case '.': case '=': case '#': case '-': case '+': case '\\': case '/': case ':': case '~':
    case '@': case '?': case '>': case '*': case '$':
      return 1;
    case '%':
      return (i + 2 < n && s[i + 1] >= '0' && s[i + 1] <= '9' && s[i + 2] >= '0' && s[i + 2] <= '9') ? 3 : 0;
    default:
      return (c >= '0' && c <= '9') ? 1 : 0;
  }
}

template <class F>
inline void split(const char* s, size_t n, F&& emit) {
  for (size_t i = 0; i < n;) {
    const size_t len = match(s, n, i);
    if (len) { emit(s + i, len); i += len; }
    else ++i;
  }
}

}  // namespace ttxtok
