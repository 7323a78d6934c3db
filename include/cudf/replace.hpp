// cudf/replace.hpp -- the part of the replace API the groupby path needs
// (reference: cpp/include/cudf/replace.hpp:29-33: replace_policy of replace_nulls / groupby::replace_nulls).
#pragma once

namespace cudf {

// which neighbouring valid value replaces a null
enum class replace_policy : bool { PRECEDING, FOLLOWING };

}  // namespace cudf
