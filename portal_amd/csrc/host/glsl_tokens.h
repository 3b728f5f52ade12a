// glsl_tokens.h -- the lexer shared by the GLSL -> C++ rewrite (glsl_translate.cpp) and the uniform-work hoister (glsl_hoist.cpp).
#pragma once
#include <string>
#include <vector>

namespace ptl {

struct Token {
    enum Kind { Space, Comment, Ident, Number, Punct, Preproc, Raw } kind;  // Raw: text inserted by a rewrite, emitted verbatim
    std::string text;
};

// Every character of the input ends up in exactly one token (concatenating the texts gives the input back).
std::vector<Token> tokenize_glsl(const std::string& s);

}  // namespace ptl
